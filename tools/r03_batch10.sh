#!/bin/bash
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
for pp in 48 24 16 12 8; do
  DDDMR_MKF_PARTS=$pp python bench.py --workload C5M --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline > gpurun_out/r03/c5m_pp_$pp.json 2> gpurun_out/r03/c5m_pp_$pp.err
  python -c "import json; d=json.load(open('gpurun_out/r03/c5m_pp_$pp.json')); m=d['config']['marking']; print('splat parts $pp', d['ms_per_step'], m['serial_schedule_ms_per_step'], m['clear_ms'], m['mark_ms'], d['config']['cmd_vel_matches_oracle'])" || tail -5 gpurun_out/r03/c5m_pp_$pp.err
done
