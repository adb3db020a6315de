"""Summarise rocprofv3 CSV output of tools/profile.sh into profiles/<tag>_*.
usage: python tools/summarize_profile.py <tag> [workload]"""
import csv, glob, json, os, sys
from collections import defaultdict

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "C2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

def short(name):
    n = name.split("(")[0]
    n = n.replace("dddmr::", "").replace("void ", "").strip()
    return n.split("<")[0] if n.startswith("k_") else n

# ---- kernel stats ----
rows = []
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
stats_out = os.path.join(dst, f"{tag}_kernel_stats.csv")
if rows:
    with open(stats_out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
    print(open(stats_out).read())

# ---- per-kernel trace: VGPR, LDS, grid ----
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    seen = {}
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if n not in seen:
            seen[n] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    json.dump(seen, open(os.path.join(dst, f"{tag}_kernel_resources.json"), "w"), indent=1)
    print(json.dumps(seen, indent=1))

# ---- PMC: average per dispatch per kernel ----
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
for k, d in summ.items():
    d["_dispatches"] = max(len(v) for v in pmc[k].values())
json.dump(summ, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(summ.get("k_score", {}), indent=1, sort_keys=True))
# HBM traffic per launch of the dominant kernel, corrected as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
# of the bytes of wide coalesced reads (x2; uncalibrated for narrower accesses).
ks = summ.get("k_score", {})
if "FETCH_SIZE" in ks and "WRITE_SIZE" in ks:
    traffic = {"k_score_hbm_bytes_per_launch": int(ks["FETCH_SIZE"] * 1024 * 2 + ks["WRITE_SIZE"] * 1024),
               "fetch_kib_raw": ks["FETCH_SIZE"], "write_kib_raw": ks["WRITE_SIZE"],
               "note": "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE as is; separate --pmc passes"}
    tp = os.path.join(dst, "traffic.json")
    allt = json.load(open(tp)) if os.path.exists(tp) else {}
    allt[workload] = traffic
    json.dump(allt, open(tp, "w"), indent=1)
    print("traffic", traffic)
