#!/bin/bash
# round-3 final measurement pass, part A: the GPU suite, smoke, the driver's bench command and a bench line per workload
set -o pipefail
O=gpurun_out/r03f; mkdir -p $O
export PYTHONPATH=$PWD
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; tail -3 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_C3_driver.json 2> $O/bench_C3_driver.err || { tail -20 $O/bench_C3_driver.err; exit 1; }
python bench.py --steps 400 --warmup 50 > $O/bench_C3.json 2> $O/bench_C3.err || { tail -20 $O/bench_C3.err; exit 1; }
python bench.py --steps 400 --warmup 50 --inputs moving --no-ceiling > $O/bench_C3_moving.json 2> $O/bench_C3_moving.err || exit 1
for W in C2 C4; do python bench.py --workload $W --steps 400 --warmup 50 > $O/bench_$W.json 2> $O/bench_$W.err || exit 1; done
python bench.py --workload shipped --no-ceiling > $O/bench_shipped.json 2> $O/bench_shipped.err || { tail -5 $O/bench_shipped.err; }
for W in C5 C5M; do python bench.py --workload $W --steps 300 --no-ceiling > $O/bench_$W.json 2> $O/bench_$W.err || exit 1; done
DDDMR_MARKING_ROUTE=general python bench.py --workload C5M --steps 300 --no-ceiling --no-cpu-baseline > $O/bench_C5M_general.json 2> $O/bench_C5M_general.err || exit 1
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 200 --no-ceiling > $O/bench_C4_2rank_gloo.json 2> $O/bench_C4_2rank_gloo.err || { tail -20 $O/bench_C4_2rank_gloo.err; }
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline())
except Exception as e:
    print(sys.argv[1], "unreadable", e); sys.exit(0)
r=d.get('roofline') or {}; c=d['config']
print(sys.argv[1].split('/')[-1], "%.2fM traj/s" % (d['value']/1e6), "ms/step", d['ms_per_step'], "k_score", r.get('kernel_ms'), "frac", r.get('frac'), "alg_tick", r.get('frac_alg_tick'), "alg_k", r.get('frac_alg_kernel'),
      "match", c.get('cmd_vel_matches_oracle'), "cpu", (d.get('cpu_baseline') or {}).get('value'), "mark", (d.get('roofline_marking') or {}).get('update_ms'), "ranks", c.get('ranks_seen') and len(c['ranks_seen']))
PY
done
