#!/bin/bash
# A/B of library builds on ONE box (box-to-box spread is +-3 %): usage tools/r03_ab.sh "<lib> <lib> ..." "<workload> ..." [rounds]
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
LIBS="$1"; WLS="$2"; R=${3:-3}
for ((i = 0; i < R; ++i)); do
  for w in $WLS; do
    for lib in $LIBS; do
      DDDMR_LIB_NAME=$lib python bench.py --workload $w --steps 400 --warmup 50 --no-ceiling --no-cpu-baseline --no-extras > gpurun_out/r03/ab.json 2> gpurun_out/r03/ab.err
      python -c "import json; d=json.load(open('gpurun_out/r03/ab.json')); print('$w', '$lib', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['cmd_vel_matches_oracle'])" || tail -3 gpurun_out/r03/ab.err
    done
  done
done
