#!/bin/bash
# persistent k_score workgroups (tile queue) on multi-round shards: GPU suite, then A/B on one box against
# (a) the same build with DDDMR_NO_PERSIST=1 and (b) the previous build (libdddmr_rollout_base.so, if present)
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
mkdir -p gpurun_out/exp_persist
run() {  # tag, workload
  timeout -k 10 120 python bench.py --workload $2 --steps 300 --no-cpu-baseline --no-ceiling > gpurun_out/exp_persist/$2_$1.json 2> gpurun_out/exp_persist/$2_$1.err || { tail -5 gpurun_out/exp_persist/$2_$1.err; return 1; }
  python - gpurun_out/exp_persist/$2_$1.json $2 $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d['roofline']
print(sys.argv[2], sys.argv[3], "ms/step", d['ms_per_step'], "k_score", r['kernel_ms'], "M/s %.1f" % (d['value']/1e6), "match", d['config']['cmd_vel_matches_oracle'])
PY
}
for rep in 1 2; do
  for W in C3 C4 C2; do
    run persist $W || exit 1
    DDDMR_NO_PERSIST=1 run nopersist $W || exit 1
    if [ -f dddmr_navigation_amd/csrc/libdddmr_rollout_base.so ]; then DDDMR_LIB_NAME=libdddmr_rollout_base.so run base $W || exit 1; fi
  done
done
