"""The instruction sequence the in-launch hand-offs rely on (csrc/rollout_kernels.hip.h, note above kBinPer), read back from
the compiler's output: in every k_score instantiation the four winner-slot stores are `global_store_dwordx2 ... sc1`
(written through past the XCD's L2), followed by `s_waitcnt vmcnt(0)` and the ticket's `global_atomic_add_u32 ... sc1`,
with no L2 write-back / invalidate (`buffer_wbl2`, `buffer_inv`) between them (the kernel's only fences are the
system-scope ones around the result record the last wave writes to host-mapped memory); the binning ticket likewise.
Prints the excerpts with the toolchain version (the text committed as profiles/r03_handoff.txt); exits non-zero when the
pattern is not found -- run it after any ROCm update.
usage: python tools/check_handoff_isa.py [extra hipcc -D flags...]"""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "dddmr_navigation_amd", "csrc", "rollout_engine.hip")
flags = ["-DDDDMR_SCORE_WPE=4", "-DDDDMR_ITEM=16", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S"] + sys.argv[1:]
ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.strip().split("\n")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "eng.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-o", out, src], stderr=subprocess.DEVNULL)
    text = open(out).read()
print("toolchain:", " | ".join(v for v in ver if "HIP version" in v or "clang version" in v))
print("flags:", " ".join(flags))
ok = True
funcs = re.findall(r"^(_ZN5dddmr(?:7k_scoreILi\d+ELb[01]ELb[01]EE|11k_bin_countE)\w*):[^\n]*\n(.*?)\n\.Lfunc_end", text, flags=re.S | re.M)
for name, body in funcs:
    lines = body.split("\n")
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    fences = [l.strip() for l in lines if re.search(r"\b(buffer_wbl2|buffer_inv)\b", l)]
    atomics = [i for i, l in enumerate(lines) if re.search(r"global_atomic_add(_u32)?\b.*\bsc1\b", l) or re.search(r"global_atomic_add(_u32)?\b", l)]
    print(f"\n== {dem}: {len(lines)} instructions, L2 write-back / invalidate instructions: {len(fences)} {fences[:4]}")
    if "k_score" in dem:
        # the ticket: the last returning global atomic add; before it the wait and the sc1 slot stores
        tick = [i for i in atomics if "sc0" in lines[i] or "glc" in lines[i]] or atomics
        t = tick[-1]
        lo = max(0, t - 40)
        ex = [l.strip() for l in lines[lo:t + 1] if re.search(r"global_store|s_waitcnt vmcnt\(0\)|global_atomic", l)]
        stores = [l for l in ex if "global_store_dwordx2" in l]
        first_store = max(i for i in range(lo, t) if "global_store_dwordx2" in lines[i]) - 3
        between = [l.strip() for l in lines[first_store:t + 1] if re.search(r"\b(buffer_wbl2|buffer_inv)\b", l)]
        good = len(stores) >= 4 and all("sc1" in l for l in stores[-4:]) and any("s_waitcnt vmcnt(0)" in l for l in ex) and not between
        print(f"   (fences of the kernel: system-scope publication of the result to host-mapped memory by the last wave; between the slot stores and the ticket: {len(between)})")
        print("   " + "\n   ".join(ex[-8:]))
        print("   pattern", "OK" if good else "NOT FOUND")
        ok &= good
    else:
        ex = [l.strip() for l in lines if re.search(r"global_atomic_add", l)]
        print("   " + "\n   ".join(ex[:6]))
        good = not fences
        print("   pattern", "OK" if good else "NOT FOUND")
        ok &= good
if not funcs:
    ok = False
    print("no k_score / k_bin_count found in the assembly")
sys.exit(0 if ok or "-DDDDMR_HANDOFF_ACQREL" in sys.argv else 1)
