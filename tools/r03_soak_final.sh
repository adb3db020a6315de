#!/bin/bash
# final-code soak of the fused marking route: chunks of 450 fresh random sequences (default generator) + one wild chunk
# usage: tools/r03_soak_final.sh [first chunk = 40] [chunks = 7]
C0=${1:-40}; NC=${2:-7}
for ((ch = C0; ch < C0 + NC; ++ch)); do bash tools/r03_soak_marking.sh $ch 450 0 || exit 1; done
bash tools/r03_soak_marking.sh $((C0 + NC)) 300 1 || exit 1
python - $C0 $NC <<'PY'
import json, sys
c0, nc = int(sys.argv[1]), int(sys.argv[2])
tot = {"sequences": 0, "updates_compared": 0, "sequences_stopped_at_a_fragile_decision": 0, "smallest_margin_of_a_stop": None}
for f in [f"gpurun_out/r03/soak_mk_0_{ch}.json" for ch in range(c0, c0 + nc)] + [f"gpurun_out/r03/soak_mk_1_{c0 + nc}.json"]:
    d = json.load(open(f))
    for k in ("sequences", "updates_compared", "sequences_stopped_at_a_fragile_decision"):
        tot[k] += d[k]
    m = d["smallest_margin_of_a_stop"]
    if m is not None:
        tot["smallest_margin_of_a_stop"] = m if tot["smallest_margin_of_a_stop"] is None else min(m, tot["smallest_margin_of_a_stop"])
tot["note"] = f"fused route, final round-3 code, {nc} chunks of 450 default sequences + 300 wild ones, seed bases 200000 + chunk * 450 from chunk {c0}"
json.dump(tot, open("gpurun_out/r03/soak_mk_final.json", "w"), indent=1)
print(tot)
PY
