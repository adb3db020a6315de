#!/bin/bash
# final-code soak of the fused marking route: 7 chunks of 450 fresh random sequences (default generator)
for ch in 30 31 32 33 34 35 36; do bash tools/r03_soak_marking.sh $ch 450 0 || exit 1; done
python - <<'PY'
import json, glob
tot = {"sequences": 0, "updates_compared": 0, "sequences_stopped_at_a_fragile_decision": 0, "smallest_margin_of_a_stop": None}
for ch in range(30, 37):
    d = json.load(open(f"gpurun_out/r03/soak_mk_0_{ch}.json"))
    for k in ("sequences", "updates_compared", "sequences_stopped_at_a_fragile_decision"):
        tot[k] += d[k]
    m = d["smallest_margin_of_a_stop"]
    if m is not None:
        tot["smallest_margin_of_a_stop"] = m if tot["smallest_margin_of_a_stop"] is None else min(m, tot["smallest_margin_of_a_stop"])
tot["note"] = "fused route, final round-3 code (LDS grid build, rank sort, 5 launches), seeds 213500..216649"
json.dump(tot, open("gpurun_out/r03/soak_mk_final.json", "w"), indent=1)
print(tot)
PY
