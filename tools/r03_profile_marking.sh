#!/bin/bash
# rocprofv3 kernel trace + stats of the C5M workload (feed + marking/clearing update + tick per step)
# usage: tools/r03_profile_marking.sh [route] [tag]
ROUTE=${1:-fused}; TAG=${2:-r03_C5M_$ROUTE}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
export PYTHONPATH=$ROOT TMPDIR=/tmp DDDMR_MARKING_ROUTE=$ROUTE; cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --workload C5M --steps 100 --warmup 20 --no-cpu-baseline --no-ceiling > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
cd $ROOT
python3 - "$TAG" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
for f in glob.glob(f"gpurun_out/prof_{tag}/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    lines = ["kernel,calls,total_ns,avg_ns,pct"]
    for r in rows[:45]:
        n = r["Name"].split("(")[0].replace("dddmr::", "").replace("void ", "")
        lines.append(",".join([n[:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]]))
    open(f"gpurun_out/{tag}_kernel_stats.csv", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))
PY
