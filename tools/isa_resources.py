"""Register / LDS / scratch usage of every kernel from the compiler's own metadata (the authoritative numbers:
rocprofv3's VGPR_Count column reports k_score's 121 VGPRs as 64).
usage: python tools/isa_resources.py > profiles/<round>_isa_resources.json"""
import json, os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "dddmr_navigation_amd", "csrc", "rollout_engine.hip")
flags = ["-DDDDMR_SCORE_WPE=4", "-DDDDMR_ITEM=16", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S"]
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "eng.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-o", out, src], stderr=subprocess.DEVNULL)
    text = open(out).read()
res = {}
for blk in re.split(r"\n\s*- \.agpr_count:", text)[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name:
        continue
    d = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", d).replace("dddmr::", "").replace("void ", "")
    if short.startswith("rocprim") or short.startswith("(anonymous"):
        continue
    get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    res[short] = {"vgpr": get("vgpr_count"), "sgpr": get("sgpr_count"), "vgpr_spills": get("vgpr_spill_count"),
                  "scratch_bytes": get("private_segment_fixed_size"), "static_lds_bytes": get("group_segment_fixed_size"),
                  "max_flat_workgroup_size": get("max_flat_workgroup_size")}
json.dump(res, sys.stdout, indent=1, sort_keys=True)
