#!/bin/bash
# the random suite under every forced launch shape / hand-off path (the default sweep only ever takes the shapes its
# small scenarios select): 1500 seeds x 2 stacks each, wild generator
export DDDMR_RANDOM_WILD=1 DDDMR_SEED_BASE=700000 DDDMR_RANDOM_SEEDS=1500
mkdir -p gpurun_out/forced
for MODE in "DDDMR_FINAL=1" "DDDMR_THREADS=256" "DDDMR_PROBE=0" "DDDMR_PROBE=1" "DDDMR_NO_ASSIGN=1" "DDDMR_RT=7" "DDDMR_NO_TAB=1" "DDDMR_FINAL=1 DDDMR_THREADS=256 DDDMR_PROBE=1"; do
  tag=$(echo $MODE | tr ' =' '__')
  env $MODE timeout -k 10 400 python -m pytest tests/test_random_gpu.py -q -m gpu -x -k "test_random_scenario and not sharded and not debug" > gpurun_out/forced/$tag.log 2>&1
  echo "$MODE: $(tail -1 gpurun_out/forced/$tag.log)"
done
