#!/bin/bash
# Kernel trace + stats only (no PMC passes): tools/trace_only.sh <tag> <python script + args...>
set -o pipefail
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
export PYTHONPATH=$ROOT
export TMPDIR=/tmp
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/"$@" > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; }
cd $ROOT
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
