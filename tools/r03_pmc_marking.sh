#!/bin/bash
# PMC passes (separate from the kernel trace) over the C5M workload; prints per-kernel averages of the marking kernels
# usage: tools/r03_pmc_marking.sh [route] [tag]
ROUTE=${1:-fused}; TAG=${2:-r03_C5M_${ROUTE}_pmc}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
export PYTHONPATH=$ROOT TMPDIR=/tmp DDDMR_MARKING_ROUTE=$ROUTE; cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY"
P2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"   # (TCC_ATOMIC / WRITE / READ_sum abort rocprofv3 on this image)
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 5 240 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o pmc -- python3 $ROOT/bench.py --workload C5M --steps 40 --warmup 10 --no-cpu-baseline --no-ceiling > $OUT/pmc$i.log 2>&1 || { echo pmc$i failed; tail -3 $OUT/pmc$i.log; }
  i=$((i+1))
done
cd $ROOT
python3 - "$TAG" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
tag = sys.argv[1]
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"gpurun_out/prof_{tag}/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("dddmr::", "").replace("void ", "").strip()
        pmc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items() if k.startswith("k_mk") or k.startswith("k_feed")}
json.dump(summ, open(f"gpurun_out/{tag}.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k[:34]:34s} waves {d.get('SQ_WAVES',0):8.0f} valu {d.get('SQ_INSTS_VALU',0):10.0f} salu {d.get('SQ_INSTS_SALU',0):9.0f} vmrd {d.get('SQ_INSTS_VMEM_RD',0):8.0f} vmwr {d.get('SQ_INSTS_VMEM_WR',0):8.0f} "
          f"busy_cyc {d.get('SQ_BUSY_CYCLES',0):9.0f} wait_any/wave_cyc {d.get('SQ_WAIT_ANY',0)/wc:5.2f} valu_act/wave_cyc {d.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} "
          f"L2 req {d.get('TCC_REQ_sum',0):9.0f} atom {d.get('TCC_ATOMIC_sum',0):8.0f} wr {d.get('TCC_WRITE_sum',0):8.0f} hit {d.get('TCC_HIT_sum',0):9.0f} miss {d.get('TCC_MISS_sum',0):8.0f}")
PY
