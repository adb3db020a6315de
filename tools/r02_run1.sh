#!/bin/bash
# round-2 GPU pass 1: parity suite, then the bench lines (new default C3, C2, C4; round-1 scene
# layouts for comparison), a 2-rank gloo rehearsal of the self-launching multi-GPU path
set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gputests.log 2>&1 || { tail -30 gpurun_out/r02/gputests.log; exit 1; }
tail -3 gpurun_out/r02/gputests.log
python bench.py > gpurun_out/r02/bench_C3.json 2> gpurun_out/r02/bench_C3.err || { tail -20 gpurun_out/r02/bench_C3.err; exit 1; }
cat gpurun_out/r02/bench_C3.json
for W in C2 C4; do
  python bench.py --workload $W > gpurun_out/r02/bench_$W.json 2> gpurun_out/r02/bench_$W.err || { tail -20 gpurun_out/r02/bench_$W.err; exit 1; }
  cat gpurun_out/r02/bench_$W.json
done
for W in C2 C3 C4; do
  python bench.py --workload $W --scene-layout r01 --no-cpu-baseline --no-ceiling > gpurun_out/r02/bench_${W}_r01.json 2> gpurun_out/r02/bench_${W}_r01.err || { tail -20 gpurun_out/r02/bench_${W}_r01.err; exit 1; }
  cat gpurun_out/r02/bench_${W}_r01.json
done
python bench.py --workload C5 --steps 300 > gpurun_out/r02/bench_C5.json 2> gpurun_out/r02/bench_C5.err || { tail -20 gpurun_out/r02/bench_C5.err; exit 1; }
cat gpurun_out/r02/bench_C5.json
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 200 > gpurun_out/r02/bench_2rank_gloo.json 2> gpurun_out/r02/bench_2rank_gloo.err || { tail -20 gpurun_out/r02/bench_2rank_gloo.err; exit 1; }
cat gpurun_out/r02/bench_2rank_gloo.json
