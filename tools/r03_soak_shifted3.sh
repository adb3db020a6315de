#!/bin/bash
# wild + scaled tick scenarios shifted by kilometres, path_blocked queries out there
mkdir -p gpurun_out/r03
export DDDMR_RANDOM_SHIFT="2750.25,-1900.5,45" DDDMR_SEED_BASE=900000 DDDMR_RANDOM_WILD=1 DDDMR_RANDOM_SCALE=2
export DDDMR_RANDOM_SEEDS=400 DDDMR_RANDOM_SHARD_SEEDS=40 DDDMR_RANDOM_SEQ_SEEDS=150 DDDMR_RANDOM_DEBUG_SEEDS=40 DDDMR_BLOCKED_CASES=300
timeout -k 10 1000 python -m pytest tests/test_random_gpu.py tests/test_path_blocked_gpu.py -q -m gpu -p no:cacheprovider -k "random" > gpurun_out/r03/soak_shifted3.log 2>&1
echo rc=$?; tail -5 gpurun_out/r03/soak_shifted3.log | cut -c1-600
cp gpurun_out/parity_stats_random.json gpurun_out/r03/soak_shifted3_random.json 2>/dev/null; cat gpurun_out/r03/soak_shifted3_random.json
