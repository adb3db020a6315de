#!/bin/bash
# extra PMC passes for k_score memory-path analysis: tools/pmc_extra.sh <tag> <script args...>
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcx_$TAG; mkdir -p $OUT
export PYTHONPATH=$ROOT TMPDIR=/tmp; cd /tmp
i=1
for P in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  echo "pass $i: $P"
  timeout -k 5 120 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/"$@" > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
  i=$((i+1))
done
cd $ROOT
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_score" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"{k:40s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
