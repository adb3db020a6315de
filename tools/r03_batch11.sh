#!/bin/bash
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
timeout -k 10 600 python -m pytest tests/test_marking_gpu.py -x -q -m gpu > gpurun_out/r03/mark11.log 2>&1; tail -2 gpurun_out/r03/mark11.log
DDDMR_DEBUG_GRID=1 python bench.py --workload C5M --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline > gpurun_out/r03/c5m_b11.json 2> gpurun_out/r03/c5m_b11.err
grep "marking update" gpurun_out/r03/c5m_b11.err | head -2
python -c "import json; d=json.load(open('gpurun_out/r03/c5m_b11.json')); m=d['config']['marking']; print('C5M', d['ms_per_step'], m['serial_schedule_ms_per_step'], m['clear_ms'], m['mark_ms'], d['config']['cmd_vel_matches_oracle'])" || tail -5 gpurun_out/r03/c5m_b11.err
for u in 4 8 16; do
  DDDMR_MKF_UNPARTS=$u python bench.py --workload C5M --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline > gpurun_out/r03/c5m_un2_$u.json 2> /dev/null
  python -c "import json; d=json.load(open('gpurun_out/r03/c5m_un2_$u.json')); m=d['config']['marking']; print('unmark parts $u', d['ms_per_step'], m['serial_schedule_ms_per_step'], m['clear_ms'], m['mark_ms'])"
done
bash tools/r03_profile_marking.sh fused r03_C5M_fused > gpurun_out/r03/prof11.log 2>&1; head -7 gpurun_out/r03/prof11.log
