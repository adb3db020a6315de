#!/bin/bash
mkdir -p gpurun_out/r03
export DDDMR_RANDOM_SHIFT="-4200.5,3100.25,-12" DDDMR_SEED_BASE=600000
export DDDMR_RANDOM_SEEDS=400 DDDMR_RANDOM_SHARD_SEEDS=40 DDDMR_RANDOM_SEQ_SEEDS=150 DDDMR_RANDOM_DEBUG_SEEDS=40 DDDMR_MARKING_SEEDS=200 DDDMR_FEED_SEEDS=150
timeout -k 10 1000 python -m pytest tests/test_random_gpu.py tests/test_marking_gpu.py tests/test_feed_gpu.py -q -m gpu -p no:cacheprovider -k "random" > gpurun_out/r03/soak_shifted2.log 2>&1
echo rc=$?; tail -8 gpurun_out/r03/soak_shifted2.log | cut -c1-600
cp gpurun_out/parity_stats_random.json gpurun_out/r03/soak_shifted2_random.json 2>/dev/null; cp gpurun_out/parity_stats_marking.json gpurun_out/r03/soak_shifted2_marking.json 2>/dev/null
cat gpurun_out/r03/soak_shifted2_random.json gpurun_out/r03/soak_shifted2_marking.json 2>/dev/null
