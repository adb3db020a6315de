#!/bin/bash
run() { python bench.py --workload $1 --steps 400 --no-cpu-baseline --no-ceiling 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%.1f M traj/s  ms/step %.5f  k_score %.5f tick_dev %.5f  rest %.5f' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms'], r['tick_device_ms'], r['tick_device_ms']-r['kernel_ms']))"; }
for W in C3 C4; do for F in 0 1 0 1; do echo -n "$W final=$F: "; DDDMR_FINAL=$F run $W; done; done
