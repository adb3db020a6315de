#!/bin/bash
# experiment: k_score compiled for 6 waves per SIMD (80 VGPRs, spills) with tiles whose LDS fits three times into a CU
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
run() {  # lib workload tile
  local tag="$1_$2_t$3"
  if [ "$3" = "0" ]; then
    DDDMR_DEBUG_GRID=1 DDDMR_LIB_NAME=$1 python bench.py --workload $2 --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline --no-extras > gpurun_out/r03/wpe_$tag.json 2> gpurun_out/r03/wpe_$tag.err
  else
    DDDMR_DEBUG_GRID=1 DDDMR_TILE=$3 DDDMR_THREADS=512 DDDMR_LIB_NAME=$1 python bench.py --workload $2 --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline --no-extras > gpurun_out/r03/wpe_$tag.json 2> gpurun_out/r03/wpe_$tag.err
  fi
  python -c "import json; d=json.load(open('gpurun_out/r03/wpe_$tag.json')); print('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['cmd_vel_matches_oracle'])" || tail -3 gpurun_out/r03/wpe_$tag.err
  grep "k_score shape" gpurun_out/r03/wpe_$tag.err | head -1
}
for lib in libdddmr_rollout.so libdddmr_rollout_wpe6.so; do
  for t in 0 4 5; do run $lib C3 $t; done
  for t in 0 5 6; do run $lib C2 $t; done
done
run libdddmr_rollout_wpe5.so C3 0
run libdddmr_rollout_wpe5.so C2 0
