#!/bin/bash
# round-2 profile pass: rocprofv3 kernel trace + stats and the separate PMC passes for one workload
# usage: tools/r02_profile.sh C3   (outputs under gpurun_out/prof_r02_<WL>/, summarised into profiles/)
WL=$1
bash tools/profile.sh r02_$WL bench.py --workload $WL --steps 120 --warmup 30 --no-cpu-baseline --no-ceiling
python3 tools/summarize_profile.py r02_$WL $WL r02 > gpurun_out/prof_r02_$WL/summary.txt 2>&1
tail -25 gpurun_out/prof_r02_$WL/summary.txt
