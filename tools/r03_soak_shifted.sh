#!/bin/bash
# random suites with the whole scenario shifted by kilometres (DDDMR_RANDOM_SHIFT): tick scenarios / shards / sequences and
# marking sequences on both routes
mkdir -p gpurun_out/r03
export DDDMR_RANDOM_SHIFT=${1:-"1500,-800,30"} DDDMR_SEED_BASE=${2:-500000}
export DDDMR_RANDOM_SEEDS=600 DDDMR_RANDOM_SHARD_SEEDS=60 DDDMR_RANDOM_SEQ_SEEDS=200 DDDMR_RANDOM_DEBUG_SEEDS=60 DDDMR_MARKING_SEEDS=250
timeout -k 10 1000 python -m pytest tests/test_random_gpu.py tests/test_marking_gpu.py -q -m gpu -p no:cacheprovider -k "random" > gpurun_out/r03/soak_shifted.log 2>&1
echo rc=$?; tail -12 gpurun_out/r03/soak_shifted.log | cut -c1-400
cp gpurun_out/parity_stats_random.json gpurun_out/r03/soak_shifted_random.json 2>/dev/null; cp gpurun_out/parity_stats_marking.json gpurun_out/r03/soak_shifted_marking.json 2>/dev/null
cat gpurun_out/r03/soak_shifted_random.json gpurun_out/r03/soak_shifted_marking.json 2>/dev/null
