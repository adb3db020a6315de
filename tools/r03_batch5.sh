#!/bin/bash
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
timeout -k 10 600 python -m pytest tests/test_marking_gpu.py -x -q -m gpu > gpurun_out/r03/mark5.log 2>&1; tail -3 gpurun_out/r03/mark5.log
for sch in overlapped serial; do
  python bench.py --workload C5M --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline --marking-schedule $sch > gpurun_out/r03/c5m_$sch.json 2> gpurun_out/r03/c5m_$sch.err
  python -c "import json; d=json.load(open('gpurun_out/r03/c5m_$sch.json')); print('$sch', d['ms_per_step'], d['value'], d['config']['marking'], d['config']['cmd_vel_matches_oracle'])" || tail -5 gpurun_out/r03/c5m_$sch.err
done
