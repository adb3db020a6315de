#!/bin/bash
mkdir -p gpurun_out/r02
run() { python bench.py --workload $1 --steps 300 --no-cpu-baseline --no-ceiling 2>gpurun_out/r02/exp5.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f M traj/s  ms/step %.5f  k_score %.5f tick_dev %.5f match %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['tick_device_ms'], d['config']['cmd_vel_matches_oracle']))"; grep "k_score shape" gpurun_out/r02/exp5.err | head -1; }
for W in C2 C3 C4; do echo -n "$W auto: "; DDDMR_DEBUG_GRID=1 run $W; done
for T in 8 10 12; do echo -n "C3 1024 tile=$T: "; DDDMR_THREADS=1024 DDDMR_TILE=$T DDDMR_DEBUG_GRID=1 run C3; done
for T in 12 16; do echo -n "C4 1024 tile=$T: "; DDDMR_THREADS=1024 DDDMR_TILE=$T DDDMR_DEBUG_GRID=1 run C4; done
for T in 16; do echo -n "C2 1024 tile=$T: "; DDDMR_THREADS=1024 DDDMR_TILE=$T DDDMR_DEBUG_GRID=1 run C2; done
