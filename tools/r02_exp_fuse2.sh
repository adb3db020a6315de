#!/bin/bash
# A/B on one box: scatter workgroups inside the k_bin_count launch (1) or as their own launch (0)
mkdir -p gpurun_out/exp_fuse
for F in 0 1 0 1; do
 for W in C2 C3; do
  DDDMR_FUSE_SCATTER=$F timeout -k 10 120 python bench.py --workload $W --steps 300 --no-cpu-baseline --no-ceiling > gpurun_out/exp_fuse/${W}_f$F.json 2> gpurun_out/exp_fuse/${W}_f$F.err || exit 1
  python - gpurun_out/exp_fuse/${W}_f$F.json $W $F <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d['roofline']
print(sys.argv[2], "fused", sys.argv[3], "ms/step", d['ms_per_step'], "k_score", r['kernel_ms'], "tick_dev", r['tick_device_ms'], "M/s %.1f" % (d['value']/1e6))
PY
 done
done
export PYTHONPATH=$PWD TMPDIR=/tmp R=$PWD; cd /tmp
for F in 0 1; do
  DDDMR_FUSE_SCATTER=$F timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/exp_fuse/tr/f$F -o t -- python3 $R/bench.py --workload C2 --steps 120 --warmup 30 --no-cpu-baseline --no-ceiling > $R/gpurun_out/exp_fuse/tr/f$F.log 2>&1
  echo "fused=$F"; find $R/gpurun_out/exp_fuse/tr/f$F -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-40,300-
done
