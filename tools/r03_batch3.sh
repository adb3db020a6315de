#!/bin/bash
# marking tests on the rebuilt library, then the C5M line and the kernel profile
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
timeout -k 10 600 python -m pytest tests/test_marking_gpu.py -x -q -m gpu > gpurun_out/r03/mark4.log 2>&1; tail -3 gpurun_out/r03/mark4.log
python bench.py --workload C5M --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline --no-extras > gpurun_out/r03/c5m_now.json 2> gpurun_out/r03/c5m_now.err
python -c "import json; d=json.load(open('gpurun_out/r03/c5m_now.json')); print('C5M', d['ms_per_step'], d['roofline_marking']['update_ms'])" || tail -3 gpurun_out/r03/c5m_now.err
bash tools/r03_profile_marking.sh fused r03_C5M_fused
