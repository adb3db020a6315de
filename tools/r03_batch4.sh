#!/bin/bash
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
for u in 48 16 8 4 2 1; do
  DDDMR_MKF_UNPARTS=$u python bench.py --workload C5M --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline --no-extras > gpurun_out/r03/c5m_un_$u.json 2> gpurun_out/r03/c5m_un_$u.err
  python -c "import json; d=json.load(open('gpurun_out/r03/c5m_un_$u.json')); print('unparts $u', d['ms_per_step'], d['roofline_marking']['update_ms'], d['config'].get('marking_matches_oracle'))" || tail -3 gpurun_out/r03/c5m_un_$u.err
done
