#!/bin/bash
# after the 16-byte rollout slots + LDS-aware rows-per-workgroup rule: parity tests that touch the rollout, then ticks
python -m pytest tests/test_parity_gpu.py tests/test_random_gpu.py tests/test_argmin_stack_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
mkdir -p gpurun_out/exp_rt
for W in C3 C4 C2; do
  python bench.py --workload $W --steps 300 --no-cpu-baseline --no-ceiling > gpurun_out/exp_rt/${W}_new.json 2> gpurun_out/exp_rt/${W}_new.err || exit 1
  python - gpurun_out/exp_rt/${W}_new.json $W <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d['roofline']
print(sys.argv[2], "ms/step", d['ms_per_step'], "k_score", r['kernel_ms'], "M/s %.1f" % (d['value']/1e6), "match", d['config']['cmd_vel_matches_oracle'])
PY
done
export DDDMR_LIB_NAME=libdddmr_rollout_diag.so PYTHONPATH=$PWD
python tools/bin_stamps.py C3 4 && python tools/bin_stamps.py C4 16
