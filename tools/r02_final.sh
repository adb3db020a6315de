#!/bin/bash
# round-2 final measurement pass: bench lines for every workload + rocprofv3 stats / PMC for C2, C3, C4
set -o pipefail
mkdir -p gpurun_out/r02f
python bench.py > gpurun_out/r02f/bench_C3.json 2> gpurun_out/r02f/bench_C3.err || { tail -20 gpurun_out/r02f/bench_C3.err; exit 1; }
for W in C2 C4; do python bench.py --workload $W > gpurun_out/r02f/bench_$W.json 2> gpurun_out/r02f/bench_$W.err || exit 1; done
for W in C2 C3 C4; do python bench.py --workload $W --scene-layout r01 --no-cpu-baseline --no-ceiling > gpurun_out/r02f/bench_${W}_r01.json 2> gpurun_out/r02f/bench_${W}_r01.err || exit 1; done
for W in C5 C5M; do python bench.py --workload $W --steps 300 --no-ceiling > gpurun_out/r02f/bench_$W.json 2> gpurun_out/r02f/bench_$W.err || exit 1; done
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 200 --no-ceiling > gpurun_out/r02f/bench_C4_2rank_gloo.json 2> gpurun_out/r02f/bench_C4_2rank_gloo.err || { tail -20 gpurun_out/r02f/bench_C4_2rank_gloo.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --workload C2 --steps 300 --no-ceiling > gpurun_out/r02f/bench_C2_2rank_gloo.json 2> gpurun_out/r02f/bench_C2_2rank_gloo.err || exit 1
PYTHONPATH=$PWD python tools/exp_pcie.py C2 > gpurun_out/r02f/pcie.txt 2>&1; PYTHONPATH=$PWD python tools/exp_pcie.py C3 >> gpurun_out/r02f/pcie.txt 2>&1
for W in C3 C2 C4; do bash tools/r02_profile.sh $W > gpurun_out/r02f/profile_$W.log 2>&1; done
# per-workgroup phase shares and the chip-wide timeline from the diagnostic build (make -C dddmr_navigation_amd/csrc diag)
if [ -f dddmr_navigation_amd/csrc/libdddmr_rollout_diag.so ]; then
  ( export DDDMR_LIB_NAME=libdddmr_rollout_diag.so PYTHONPATH=$PWD
    for W in C2 C3 C4; do python tools/phase_stamps.py $W | grep -v "deciles\|^ [0-9]\|start times"; done
    python tools/bin_stamps.py C2 1; python tools/bin_stamps.py C3 4; python tools/bin_stamps.py C4 16 ) > gpurun_out/r02f/phase_stamps.txt 2>&1
fi
tools/ubench/f64_rate > gpurun_out/r02f/ubench_f64_rate.txt 2>&1
for f in gpurun_out/r02f/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d['roofline']; c=d['config']
print(sys.argv[1].split('/')[-1], "%.1fM traj/s" % (d['value']/1e6), "ms/step", d['ms_per_step'], "k_score", r['kernel_ms'], "tick_dev", r['tick_device_ms'], "coll", c['colliding_share'], "frac", r['frac'], "match", c['cmd_vel_matches_oracle'], "sync", c['sync_tick_latency_ms'], "cpu", (d['cpu_baseline'] or {}).get('value'))
PY
done
