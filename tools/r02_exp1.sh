#!/bin/bash
# experiment: k_finalize vs in-kernel ticket, 4 vs 5 waves per SIMD
mkdir -p gpurun_out/r02
for W in C2 C3 C4; do
  for F in 0 1; do
    for LIB in libdddmr_rollout.so libdddmr_rollout_wpe5.so; do
      echo -n "$W final=$F $LIB: "
      DDDMR_FINAL=$F DDDMR_LIB_NAME=$LIB python bench.py --workload $W --steps 300 --no-cpu-baseline --no-ceiling 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f M traj/s  ms/step %.5f  k_score %.5f  match %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['cmd_vel_matches_oracle']))"
    done
  done
done
