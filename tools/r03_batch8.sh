#!/bin/bash
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
timeout -k 10 600 python -m pytest tests/test_marking_gpu.py -x -q -m gpu > gpurun_out/r03/mark9.log 2>&1; tail -3 gpurun_out/r03/mark9.log
for u in groups roots; do
  DDDMR_MKF_UNMARK=$u python bench.py --workload C5M --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline > gpurun_out/r03/c5m_un_$u.json 2> gpurun_out/r03/c5m_un_$u.err
  python -c "import json; d=json.load(open('gpurun_out/r03/c5m_un_$u.json')); m=d['config']['marking']; print('unmark in $u', d['ms_per_step'], m['serial_schedule_ms_per_step'], m['clear_ms'], m['mark_ms'], m['route'], d['config']['cmd_vel_matches_oracle'])" || tail -5 gpurun_out/r03/c5m_un_$u.err
done
bash tools/r03_profile_marking.sh fused r03_C5M_fused > gpurun_out/r03/prof8.log 2>&1; head -8 gpurun_out/r03/prof8.log
