#!/bin/bash
# final-code soak of the tick path: fresh seed base, the random scenarios (x2 stack orders), shards, sequences, debug
# outputs, the RCCL path with one rank and the loopback exchange, the feed and path_blocked sweeps
mkdir -p gpurun_out/r03
export DDDMR_SEED_BASE=${1:-300000} DDDMR_RANDOM_SEEDS=${2:-2500} DDDMR_RANDOM_SHARD_SEEDS=300 DDDMR_RANDOM_SEQ_SEEDS=1500 DDDMR_RANDOM_DEBUG_SEEDS=300 DDDMR_COMM_SEEDS=400
timeout -k 10 1100 python -m pytest tests/test_random_gpu.py tests/test_comm_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r03/soak_tick.log 2>&1
tail -3 gpurun_out/r03/soak_tick.log
cp gpurun_out/parity_stats_random.json gpurun_out/r03/soak_tick_stats.json 2>/dev/null; cat gpurun_out/r03/soak_tick_stats.json 2>/dev/null | head -30
