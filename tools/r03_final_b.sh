#!/bin/bash
# round-3 final measurement pass, part B: rocprofv3 kernel stats + separate PMC passes (C3, C2, C5M both routes)
mkdir -p gpurun_out/r03f
export PYTHONPATH=$PWD
for WL in C3 C2; do
  bash tools/profile.sh r03_$WL bench.py --workload $WL --steps 120 --warmup 30 --no-cpu-baseline --no-ceiling --no-extras > gpurun_out/r03f/profile_$WL.log 2>&1
  python3 tools/summarize_profile.py r03_$WL $WL r03 > gpurun_out/prof_r03_$WL/summary.txt 2>&1; tail -12 gpurun_out/prof_r03_$WL/summary.txt
done
bash tools/r03_profile_marking.sh fused r03_C5M_fused
bash tools/r03_profile_marking.sh general r03_C5M_general
bash tools/r03_pmc_marking.sh fused r03_C5M_fused_pmc
