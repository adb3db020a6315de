"""Per-workgroup phase breakdown of k_bin_count (binning / assignment / rollout workgroups) from the
diagnostic build (make -C dddmr_navigation_amd/csrc diag).  Shares, not run time, are meaningful.
usage: DDDMR_LIB_NAME=libdddmr_rollout_diag.so python tools/bin_stamps.py C3 4"""
import ctypes as C, sys
import numpy as np
from dddmr_navigation_amd import scenes, _capi as K
from dddmr_navigation_amd.local_planner import LocalPlanner

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = scenes.bench_scene(cfg)
lib = K.load_library()
with LocalPlanner([sc.theory], max_points=max(len(sc.cloud), 16)) as lp:
    lp.set_cloud(sc.cloud); lp.setPlan(sc.plan)
    name = sc.theory.name.decode()
    for _ in range(5):
        r = lp.tick(name, sc.tick)
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.dddmr_rollout_diag_rstamps.argtypes = [C.c_void_p, C.c_size_t]
    assert lib.dddmr_rollout_diag_rstamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    st = buf.reshape(4096, 8).astype(np.int64)
    n_pts = len(sc.cloud)
    per_wg = 1024 if n_pts <= 128 * 1024 else 4096
    nb = max(1, min(512, (n_pts + per_wg - 1) // per_wg))
    live = st[:, 0] > 0
    t0 = st[live, 0].min()
    ag = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # assignment workgroups (n_local / 4096, rounded up)
    # block order of k_bin_count: assignment, rollout, binning workgroups
    n_live = int(live.sum())
    asg = np.zeros(4096, bool); asg[:ag] = True
    binb = np.zeros(4096, bool); binb[n_live - nb:n_live] = True
    roll = live & ~asg & ~binb
    u = 1000.0
    def show(nm, a, b, m):
        d = (st[m, b] - st[m, a]) / u
        print(f"  {nm:34s} n {m.sum():5d} mean {d.mean():8.2f} p50 {np.percentile(d,50):8.2f} max {d.max():8.2f}")
    print(cfg, "points", n_pts, "bin wgs", nb, "assign wgs", int(asg.sum()), "rollout wgs", int(roll.sum()), "(units: 1000 s_memtime ticks)")
    show("bin: count pass", 0, 1, binb)
    show("bin: last wg cell scan", 1, 3, binb)
    if asg.any(): show("assign", 0, 3, asg)
    show("rollout A (theta chain)", 0, 1, roll)
    show("rollout B (sincos, increments)", 1, 2, roll)
    show("rollout C (xy chain + copy-out)", 2, 3, roll)
    # slots 4 / 5: s_memrealtime (100 MHz, one time base for the whole chip) at the first and last stamp
    w0 = st[live, 4].min()
    for nm, m in (("bin", binb), ("assign", asg), ("rollout", roll)):
        if m.any():
            s = (st[m, 4] - w0) / 100.0; en = (st[m, 5] - w0) / 100.0
            print(f"  {nm:8s} wall start us: min {s.min():6.2f} p50 {np.percentile(s,50):6.2f} p90 {np.percentile(s,90):6.2f} max {s.max():6.2f} | end: p50 {np.percentile(en,50):6.2f} p90 {np.percentile(en,90):6.2f} max {en.max():6.2f} | life p50 {np.percentile(en-s,50):6.2f}")
