"""Phase breakdown of the k_bin_count launch (binning / assignment / rollout workgroups)
from the diagnostic build.
usage: DDDMR_LIB_NAME=libdddmr_rollout_diag.so python tools/rollout_stamps.py C2"""
import ctypes as C, sys
import numpy as np
from dddmr_navigation_amd import scenes, _capi as K
from dddmr_navigation_amd.local_planner import LocalPlanner

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
sc = scenes.bench_scene(cfg)
lib = K.load_library()
with LocalPlanner([sc.theory], max_points=max(len(sc.cloud), 16)) as lp:
    lp.set_cloud(sc.cloud); lp.setPlan(sc.plan)
    name = sc.theory.name.decode()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.dddmr_rollout_diag_rstamps.argtypes = [C.c_void_p, C.c_size_t]
    for _ in range(5):
        r = lp.tick(name, sc.tick)
    assert lib.dddmr_rollout_diag_rstamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    st = buf.reshape(4096, 8).astype(np.int64)
    used = st[:, 3] > 0
    idx = np.nonzero(used)[0]
    st = st[used]
    t0 = st[:, 0].min()
    print(cfg, "blocks", len(st), "n_local", r.n_local)
    # classify: rollout blocks have stamps 1,2 between 0 and 3; bin blocks have stamp 1 and no 2
    for i, row in zip(idx[:14], st[:14]):
        print("  block", i, "start", (row[0]-t0)/1e3, "s1", (row[1]-row[0])/1e3, "s2", (row[2]-row[0])/1e3 if row[2] else None, "end", (row[3]-row[0])/1e3)
    roll = st[(st[:, 2] > 0)]
    if len(roll):
        print("  rollout blocks", len(roll))
        for nm, a, b in (("A chain", 0, 1), ("B trig", 1, 2), ("C chain", 2, 3)):
            d = (roll[:, b] - roll[:, a]) / 1e3
            d = d[(d > 0) & (d < 1e4)]
            print(f"    {nm:8s} n {len(d)} median {np.median(d):7.2f} kc  p90 {np.percentile(d,90):7.2f} max {d.max():7.2f}")
        d = (roll[:, 3] - roll[:, 0]) / 1e3
        d = d[(d > 0) & (d < 1e4)]
        print(f"    block    n {len(d)} median {np.median(d):7.2f} kc  p90 {np.percentile(d,90):7.2f} max {d.max():7.2f}")
        print("    start: min %.1f max %.1f kc; end max %.1f kc" % ((roll[:,0].min()-t0)/1e3, (roll[:,0].max()-t0)/1e3, (roll[:,3].max()-t0)/1e3))
    asg = st[(st[:, 1] == 0) & (st[:, 2] == 0)]
    d = (asg[:, 3] - asg[:, 0]) / 1e3
    print("  assignment blocks", len(asg), "durations kc", np.round(d, 2))
    binb = st[(st[:, 1] > 0) & (st[:, 2] == 0)]
    d = (binb[:, 3] - binb[:, 0]) / 1e3
    print("  binning blocks", len(binb), "median %.2f max %.2f kc" % (np.median(d), d.max()))
    print("  launch span %.1f kc" % ((st[:, 3].max() - t0) / 1e3))
