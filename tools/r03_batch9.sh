#!/bin/bash
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
timeout -k 10 600 python -m pytest tests/test_marking_gpu.py -x -q -m gpu > gpurun_out/r03/mark10.log 2>&1; tail -2 gpurun_out/r03/mark10.log
DDDMR_LIB_NAME=libdddmr_rollout_diag.so timeout -k 10 200 python tools/marking_stamps.py 2>&1 | grep -B1 -A16 "partition 0"
python bench.py --workload C5M --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline > gpurun_out/r03/c5m_b9.json 2> gpurun_out/r03/c5m_b9.err
python -c "import json; d=json.load(open('gpurun_out/r03/c5m_b9.json')); m=d['config']['marking']; print('C5M', d['ms_per_step'], m['serial_schedule_ms_per_step'], m['clear_ms'], m['mark_ms'], d['config']['cmd_vel_matches_oracle'])" || tail -5 gpurun_out/r03/c5m_b9.err
bash tools/r03_profile_marking.sh fused r03_C5M_fused > gpurun_out/r03/prof9.log 2>&1; head -7 gpurun_out/r03/prof9.log
