#!/bin/bash
# per-workgroup winner slots on single-round shards: whole GPU suite, then ticks and the C2 timeline
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
mkdir -p gpurun_out/exp_slots
for W in C2 C3 C4; do
  python bench.py --workload $W --steps 300 --no-cpu-baseline --no-ceiling > gpurun_out/exp_slots/${W}.json 2> gpurun_out/exp_slots/${W}.err || exit 1
  python - gpurun_out/exp_slots/${W}.json $W <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d['roofline']
print(sys.argv[2], "ms/step", d['ms_per_step'], "k_score", r['kernel_ms'], "M/s %.1f" % (d['value']/1e6), "match", d['config']['cmd_vel_matches_oracle'])
PY
done
export DDDMR_LIB_NAME=libdddmr_rollout_diag.so PYTHONPATH=$PWD
python tools/phase_stamps.py C2 | grep -E "E split|wall|resident|CUs|last workgroup"
