#!/bin/bash
# scatter workgroups fused into the k_bin_count launch: whole GPU suite under a timeout (a wait that never ends
# must not hang the box), then ticks and the k_bin_count timeline
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
mkdir -p gpurun_out/exp_fuse
for W in C2 C3 C4; do
  timeout -k 10 120 python bench.py --workload $W --steps 300 --no-cpu-baseline --no-ceiling > gpurun_out/exp_fuse/${W}.json 2> gpurun_out/exp_fuse/${W}.err || exit 1
  python - gpurun_out/exp_fuse/${W}.json $W <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d['roofline']
print(sys.argv[2], "ms/step", d['ms_per_step'], "k_score", r['kernel_ms'], "M/s %.1f" % (d['value']/1e6), "match", d['config']['cmd_vel_matches_oracle'])
PY
done
export DDDMR_LIB_NAME=libdddmr_rollout_diag.so PYTHONPATH=$PWD
timeout -k 10 60 python tools/bin_stamps.py C2 1 && timeout -k 10 60 python tools/bin_stamps.py C3 4
