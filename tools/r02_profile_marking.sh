#!/bin/bash
# rocprofv3 kernel trace + stats of the C5M workload (feed + marking/clearing update + tick per step)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r02_C5M; mkdir -p $OUT
export PYTHONPATH=$ROOT TMPDIR=/tmp; cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --workload C5M --steps 100 --warmup 20 --no-cpu-baseline --no-ceiling > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
cd $ROOT
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_r02_C5M/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("kernel,calls,total_ns,avg_ns,pct")
    for r in rows[:45]:
        n = r["Name"].split("(")[0].replace("dddmr::", "").replace("void ", "")
        print(",".join([n[:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]]))
PY
