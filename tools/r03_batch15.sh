#!/bin/bash
# k_bin_count's duration at C3 for several rollout rows per workgroup (DDDMR_RT), rocprofv3 kernel stats
mkdir -p gpurun_out/r03
ROOT=$PWD; export PYTHONPATH=$ROOT TMPDIR=/tmp
for rt in 50 43 37 32 25; do
  OUT=$ROOT/gpurun_out/r03/rt_$rt; mkdir -p $OUT
  (cd /tmp && DDDMR_RT=$rt timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $ROOT/bench.py --workload C3 --steps 150 --warmup 30 --no-cpu-baseline --no-ceiling --no-extras > $OUT/log.txt 2>&1)
  python3 - $rt <<'PY'
import csv, glob, sys
rt = sys.argv[1]
for f in glob.glob(f"gpurun_out/r03/rt_{rt}/**/*kernel_stats.csv", recursive=True):
    out = []
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("dddmr::", "").replace("void ", "")
        if n.startswith("k_bin_count") or n.startswith("k_score<512, true, false"):
            out.append(f"{n[:24]} {float(r['AverageNs']) / 1e3:.2f} us")
    print("rt", rt, " | ".join(out))
PY
done
