#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03/full3.log 2>&1; tail -3 gpurun_out/r03/full3.log
for w in C2 C3; do
  for lib in libdddmr_rollout.so libdddmr_rollout_acqrel.so; do
    DDDMR_LIB_NAME=$lib python bench.py --workload $w --steps 400 --warmup 50 --no-ceiling --no-cpu-baseline --no-extras > gpurun_out/r03/handoff_${w}_$lib.json 2> gpurun_out/r03/handoff_${w}_$lib.err
    python -c "import json; d=json.load(open('gpurun_out/r03/handoff_${w}_$lib.json')); print('$w', '$lib', d['ms_per_step'], d['roofline']['kernel_ms'])" || tail -3 gpurun_out/r03/handoff_${w}_$lib.err
  done
done
for w in C3 C4; do
  for t in 0 1; do
    DDDMR_NO_TAIL_ROUND=$t python bench.py --workload $w --steps 400 --warmup 50 --no-ceiling --no-cpu-baseline --no-extras > gpurun_out/r03/tail_${w}_$t.json 2>/dev/null
    python -c "import json; d=json.load(open('gpurun_out/r03/tail_${w}_$t.json')); print('$w no_tail_round=$t', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['cmd_vel_matches_oracle'])"
  done
done
