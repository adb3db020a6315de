#!/bin/bash
# one chunk of the randomised marking soak on the fused route: tools/r03_soak_marking.sh <chunk> [seeds per chunk = 450] [wild = 0]
CH=${1:-0}; N=${2:-450}; WILD=${3:-0}
mkdir -p gpurun_out/r03
export DDDMR_MARKING_SEEDS=$N DDDMR_SEED_BASE=$((200000 + CH * N)) DDDMR_RANDOM_WILD=$WILD
timeout -k 10 1150 python -m pytest tests/test_marking_gpu.py -x -q -m gpu -k "random_marking_sequences and fused" -p no:cacheprovider > gpurun_out/r03/soak_mk_${WILD}_$CH.log 2>&1
tail -3 gpurun_out/r03/soak_mk_${WILD}_$CH.log
cp gpurun_out/parity_stats_marking.json gpurun_out/r03/soak_mk_${WILD}_$CH.json 2>/dev/null
