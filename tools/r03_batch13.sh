#!/bin/bash
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03/suite13.log 2>&1; tail -2 gpurun_out/r03/suite13.log
for w in C3 C2 C4; do
  python bench.py --workload $w --steps 400 --warmup 50 --no-ceiling --no-cpu-baseline --no-extras > gpurun_out/r03/pf_$w.json 2> gpurun_out/r03/pf_$w.err
  python -c "import json; d=json.load(open('gpurun_out/r03/pf_$w.json')); print('$w', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['cmd_vel_matches_oracle'])" || tail -3 gpurun_out/r03/pf_$w.err
done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/pf_driver.json 2> gpurun_out/r03/pf_driver.err
python -c "import json; d=json.load(open('gpurun_out/r03/pf_driver.json')); print('driver', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['config']['cmd_vel_matches_oracle'])"
