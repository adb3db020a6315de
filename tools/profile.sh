#!/bin/bash
# Profile one workload on the GPU box: kernel trace + stats, then PMC passes.
# usage: tools/profile.sh <tag> <python script + args...>
# Outputs land under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns
# them into profiles/<tag>_*.{csv,json}.
set -o pipefail
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export PYTHONPATH=$ROOT
export TMPDIR=/tmp
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/"$@" > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; }
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY"
P2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  echo "pmc pass $i"; timeout -k 5 180 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o pmc -- python3 $ROOT/"$@" > $OUT/pmc$i.log 2>&1 || { echo pmc$i failed; tail -3 $OUT/pmc$i.log; }
  i=$((i+1))
done
cd $ROOT
find $OUT -name "*.csv" | head -30
