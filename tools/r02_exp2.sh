#!/bin/bash
mkdir -p gpurun_out/r02
run() { python bench.py --workload $1 --steps 200 --no-cpu-baseline --no-ceiling 2>gpurun_out/r02/exp2.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f M traj/s  ms/step %.5f  k_score %.5f  match %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['cmd_vel_matches_oracle']))"; }
for W in C3 C4; do
  echo -n "$W auto: "; DDDMR_DEBUG_GRID=1 run $W; grep "k_score shape" gpurun_out/r02/exp2.err | head -1
  echo -n "$W auto no_tab: "; DDDMR_NO_TAB=1 run $W
  for T in 2 3 4 5; do echo -n "$W tile=$T: "; DDDMR_TILE=$T DDDMR_DEBUG_GRID=1 run $W; grep "k_score shape" gpurun_out/r02/exp2.err | head -1; done
  echo -n "$W tile=2 no_tab: "; DDDMR_TILE=2 DDDMR_NO_TAB=1 run $W
done
