#!/bin/bash
# experiment: trajectories per rollout workgroup (DDDMR_RT) vs tick time; the rollout's LDS rows decide how many
# 1024-lane workgroups of k_bin_count fit a CU
mkdir -p gpurun_out/exp_rt
for W in C3 C4 C2; do
  for RT in 0 16 24 32 48 64; do
    if [ $RT = 0 ]; then unset DDDMR_RT; else export DDDMR_RT=$RT; fi
    python bench.py --workload $W --steps 300 --no-cpu-baseline --no-ceiling > gpurun_out/exp_rt/${W}_$RT.json 2> gpurun_out/exp_rt/${W}_$RT.err || exit 1
    python - gpurun_out/exp_rt/${W}_$RT.json $W $RT <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d['roofline']
print(sys.argv[2], "rt", sys.argv[3], "ms/step", d['ms_per_step'], "k_score", r['kernel_ms'], "M/s %.1f" % (d['value']/1e6))
PY
  done
done
