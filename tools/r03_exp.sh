#!/bin/bash
# diagnostic build only: kernel times of the fused marking update with parts of its launches left out (DDDMR_MKF_EXP bits:
# 1 no union-find blocks, 2 no ray-test blocks, 4 no commit blocks, 8 no node-by-node dGraph blocks, 16 one walk block)
export PYTHONPATH=$PWD DDDMR_LIB_NAME=libdddmr_rollout_diag.so TMPDIR=/tmp DDDMR_MARKING_ROUTE=fused
for e in "$@"; do
  mkdir -p gpurun_out/exp$e
  (cd /tmp && DDDMR_MKF_EXP=$e timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp$e -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload C5M --steps 60 --warmup 10 --no-cpu-baseline --no-ceiling > /dev/null 2>&1)
  python3 - $e <<'PY'
import csv, sys
e = sys.argv[1]
out = []
for r in csv.DictReader(open(f"gpurun_out/exp{e}/t_kernel_stats.csv")):
    if "k_mkf" in r["Name"]:
        out.append(f'{r["Name"].split("(")[0].replace("dddmr::k_mkf_", "")} {float(r["AverageNs"]) / 1e3:.1f}')
print("exp", e, " | ".join(sorted(out)))
PY
done
