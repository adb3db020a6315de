#!/bin/bash
mkdir -p gpurun_out/r03 gpurun_out/r03f
export PYTHONPATH=$PWD
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden" > gpurun_out/r03/golden14.log 2>&1; tail -4 gpurun_out/r03/golden14.log
python bench.py --workload C3P --steps 400 --warmup 50 --no-ceiling > gpurun_out/r03f/bench_C3P.json 2> gpurun_out/r03f/bench_C3P.err
python -c "import json; d=json.load(open('gpurun_out/r03f/bench_C3P.json')); print('C3P', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['config']['colliding_share'], d['config']['cmd_vel_matches_oracle'], d['cpu_baseline']['value'])" || tail -5 gpurun_out/r03f/bench_C3P.err
