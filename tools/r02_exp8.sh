#!/bin/bash
mkdir -p gpurun_out/r02
run() { python bench.py --workload $1 --steps 300 --no-cpu-baseline --no-ceiling 2>gpurun_out/r02/exp8.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f M traj/s  ms/step %.5f  k_score %.5f tick_dev %.5f match %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['tick_device_ms'], d['config']['cmd_vel_matches_oracle']))"; grep "k_score shape" gpurun_out/r02/exp8.err | head -1; }
for C in 0.2 0.25 0.3 0.35 0.42 0.5; do for W in C2 C3 C4; do echo -n "$W cell=$C: "; DDDMR_CELL=$C DDDMR_DEBUG_GRID=1 run $W; done; done
