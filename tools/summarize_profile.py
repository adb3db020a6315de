"""Summarise rocprofv3 CSV output of tools/profile.sh into profiles/<tag>_*.
usage: python tools/summarize_profile.py <tag> [workload] [round]

Besides the per-tag files it maintains three per-round tables keyed by workload, which bench.py
reads for the workload it ran: profiles/traffic.json (HBM bytes per k_score launch),
profiles/<round>_pmc.json (what binds k_score) and profiles/<round>_kernel_avg.json (rocprofv3's
average kernel durations, to be compared with bench.py's HIP-event time)."""
import csv, glob, json, os, sys
from collections import defaultdict

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "C2"
rnd = sys.argv[3] if len(sys.argv) > 3 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

def short(name, keep_template=False):
    n = name.split("(")[0]
    n = n.replace("dddmr::", "").replace("void ", "").strip()
    return n.split("<")[0] if n.startswith("k_") and not keep_template else n


def merged_avg_ns(rows):
    """calls-weighted average duration per kernel, all template instantiations of a kernel pooled
    (k_score<..., kProbe=true> runs on the first tick only, the lean instantiation on the rest)"""
    tot, calls = defaultdict(float), defaultdict(int)
    for r in rows:
        tot[short(r["Name"])] += float(r["TotalDurationNs"])
        calls[short(r["Name"])] += int(r["Calls"])
    return {k: tot[k] / calls[k] for k in tot if calls[k]}

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
            w.writerow([short(r["Name"], keep_template=True), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
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
    traffic["profile"] = tag
    allt[workload] = traffic
    json.dump(allt, open(tp, "w"), indent=1)
    print("traffic", traffic)


def update(path, key, value):
    allv = json.load(open(path)) if os.path.exists(path) else {}
    allv[key] = value
    json.dump(allv, open(path, "w"), indent=1, sort_keys=True)


# rocprofv3's average durations (ms) of the tick's kernels for this workload
if rows:
    avg = {k + "_ms": round(v * 1e-6, 6) for k, v in merged_avg_ns(rows).items() if k.startswith("k_")}
    avg["source"] = f"profiles/{tag}_kernel_stats.csv"
    update(os.path.join(dst, f"{rnd}_kernel_avg.json"), workload, avg)
    print("kernel averages", avg)

# what binds k_score: SQ_* counters count quad-cycles (MI355X_MICROARCH.md); WAIT_ANY + WAIT_INST_ANY +
# ACTIVE_INST_ANY ~ WAVE_CYCLES.  VALU busy = VALU issue cycles over the SIMD-cycles of the launch
# (1024 SIMDs x rocprofv3's average duration x 2.4 GHz peak clock: a lower bound of the share).
if "SQ_WAVE_CYCLES" in ks and rows:
    dur_ns = merged_avg_ns(rows).get("k_score")
    lim = {
        "waves_waiting_frac": round(ks["SQ_WAIT_ANY"] / ks["SQ_WAVE_CYCLES"], 4),
        "issue_stall_frac": round(ks["SQ_WAIT_INST_ANY"] / ks["SQ_WAVE_CYCLES"], 4),
        "issuing_frac": round(ks["SQ_ACTIVE_INST_ANY"] / ks["SQ_WAVE_CYCLES"], 4),
        "valu_busy_frac": round(4.0 * ks["SQ_ACTIVE_INST_VALU"] / (dur_ns * 2.4 * 1024), 4) if dur_ns else None,
        "lds_bank_conflict_per_lds_active": round(ks.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(ks.get("SQ_ACTIVE_INST_LDS", 1.0), 1.0), 3),
        "l2_hit_rate": round(ks["TCC_HIT_sum"] / max(ks["TCC_HIT_sum"] + ks["TCC_MISS_sum"], 1.0), 4) if "TCC_HIT_sum" in ks else None,
        "valu_insts_per_launch": ks.get("SQ_INSTS_VALU"), "waves_per_launch": ks.get("SQ_WAVES"),
        "k_score_avg_ns": dur_ns, "source": f"profiles/{tag}_pmc.json (separate --pmc passes)"}
    update(os.path.join(dst, f"{rnd}_pmc.json"), workload, lim)
    print("limiter", lim)
