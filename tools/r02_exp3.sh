#!/bin/bash
mkdir -p gpurun_out/r02
run() { python bench.py --workload $1 --steps 200 --no-cpu-baseline --no-ceiling 2>gpurun_out/r02/exp3.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f M traj/s  ms/step %.5f  k_score %.5f  match %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['cmd_vel_matches_oracle']))"; grep "k_score shape" gpurun_out/r02/exp3.err | head -1; }
for T in 4 5 6; do echo -n "C3 256 tile=$T: "; DDDMR_THREADS=256 DDDMR_TILE=$T DDDMR_DEBUG_GRID=1 run C3; done
for T in 6 8 10 12; do echo -n "C3 512 tile=$T: "; DDDMR_THREADS=512 DDDMR_TILE=$T DDDMR_DEBUG_GRID=1 run C3; done
for T in 5 6 7 8; do echo -n "C4 256 tile=$T: "; DDDMR_THREADS=256 DDDMR_TILE=$T DDDMR_DEBUG_GRID=1 run C4; done
for T in 8 10 12 16; do echo -n "C4 512 tile=$T: "; DDDMR_THREADS=512 DDDMR_TILE=$T DDDMR_DEBUG_GRID=1 run C4; done
