"""Per-workgroup phase breakdown of k_score from the diagnostic build
(make -C dddmr_navigation_amd/csrc diag).  Shares, not run time, are meaningful.
usage: DDDMR_LIB_NAME=libdddmr_rollout_diag.so python tools/phase_stamps.py C2"""
import ctypes as C, sys
import numpy as np
from dddmr_navigation_amd import scenes, _capi as K
from dddmr_navigation_amd.local_planner import LocalPlanner

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
sc = scenes.bench_scene(cfg)
import os
if os.environ.get("EXP_EMPTY"):
    sc.cloud = sc.cloud[:0]
lib = K.load_library()
with LocalPlanner([sc.theory], max_points=max(len(sc.cloud), 16)) as lp:
    lp.set_cloud(sc.cloud); lp.setPlan(sc.plan)
    name = sc.theory.name.decode()
    for _ in range(5):
        r = lp.tick(name, sc.tick)
    SL = 20
    n_wg = 16384
    buf = np.zeros(n_wg * SL, dtype=np.uint64)
    lib.dddmr_rollout_diag_stamps.argtypes = [C.c_void_p, C.c_size_t]
    assert lib.dddmr_rollout_diag_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    st = buf.reshape(n_wg, SL)
    st = st.astype(np.int64)
    nwg = int(os.environ.get('EXP_NWG', '16384'))
    used = np.zeros(len(st), bool); used[:nwg] = True
    used &= st[:, 7] > 0
    st = st[used]
    print(cfg, "workgroups", len(st), "score_ms", r.score_ms)
    names = ["A headers+index", "pair offsets", "(unused)", "D1 pose/obb", "D2 segments+scan", "D3 walk", "P+E path+score"]
    t0 = st[:, 0].min()
    for i, nm in enumerate(names):
        d = (st[:, i + 1] - st[:, i]) / 1000.0   # kilo-cycles (s_memtime counts shader clocks)
        print(f"  {nm:18s} mean {d.mean():8.2f} kc  p50 {np.percentile(d,50):8.2f}  p99 {np.percentile(d,99):8.2f}  max {d.max():8.2f}")
    dP = (st[:, 8] - st[:, 6]) / 1000.0; dE = (st[:, 7] - st[:, 8]) / 1000.0
    print(f"    of which P path 1-NN mean {dP.mean():8.2f} kc max {dP.max():8.2f};  E stick+score+argmin mean {dE.mean():8.2f} max {dE.max():8.2f}")
    e1 = (st[:, 10] - st[:, 8]) / 1000.0; e2 = (st[:, 11] - st[:, 10]) / 1000.0; e3 = (st[:, 7] - st[:, 11]) / 1000.0
    print(f"    E split: StickPath sums mean {e1.mean():6.2f} kc | stacked scoring + stores mean {e2.mean():6.2f} | reduce + atomics (+ ticket) mean {e3.mean():6.2f} max {e3.max():6.2f}")
    life = (st[:, 7] - st[:, 0]) / 1000.0
    print(f"  workgroup lifetime mean {life.mean():.2f} kc max {life.max():.2f} kc (s_memtime runs per XCC: only differences inside a workgroup mean anything; ~2.2 kc per us)")
    tot = st[:, 9]
    print(f"  items per wg: mean {tot.mean():.0f} max {tot.max()}  corr(items, D3 time) {np.corrcoef(tot, st[:,6]-st[:,5])[0,1]:.3f}")

    # chip-wide timeline from s_memrealtime (100 MHz): when workgroups start / end, how many are resident, per-CU share
    w0 = st[:, 12].min()
    ws = (st[:, 12] - w0) / 100.0; we = (st[:, 13] - w0) / 100.0
    span = we.max()
    print(f"  wall: first start 0, last start {ws.max():.2f} us, last end {span:.2f} us; workgroup life p50 {np.percentile(we-ws,50):.2f} p90 {np.percentile(we-ws,90):.2f} max {(we-ws).max():.2f} us")
    grid = np.linspace(0.0, span, 13)[:-1]
    res = [(int(((ws <= t) & (we > t)).sum())) for t in grid]
    print("  resident workgroups at", " ".join(f"{t:.0f}us:{n}" for t, n in zip(grid, res)))
    hw = st[:, 14]
    xcc = (hw >> 32) & 0xF; hid = hw & 0xFFFFFFFF
    cu = (xcc << 8) | (((hid >> 13) & 7) << 5) | (((hid >> 12) & 1) << 4) | ((hid >> 8) & 0xF)
    ucu, cnt = np.unique(cu, return_counts=True)
    busy = np.array([(we[cu == c] - ws[cu == c]).sum() for c in ucu])
    print(f"  CUs used {len(ucu)}; workgroups per CU min {cnt.min()} p50 {int(np.median(cnt))} max {cnt.max()}; per-CU busy (sum of lives / 2 slots) p10 {np.percentile(busy,10)/2:.1f} p50 {np.percentile(busy,50)/2:.1f} p90 {np.percentile(busy,90)/2:.1f} max {busy.max()/2:.1f} us of {span:.1f}")
    x, xc = np.unique(xcc, return_counts=True)
    print("  workgroups per XCC:", dict(zip(x.tolist(), xc.tolist())))

    # the last workgroup's hand-off (single-round shards): ticket drawn -> slots loaded -> reduced -> published
    last = int(np.argmax(st[:, 18])) if st[:, 18].max() > 0 else -1
    if last >= 0:
        t = st[last]
        print(f"  last workgroup: stores+ticket {(t[15]-t[11])/1000.0:.2f} kc | slot loads + wave reduce {(t[16]-t[15])/1000.0:.2f} | cross-wave reduce {(t[17]-t[16])/1000.0:.2f} | result + system fence + seq {(t[18]-t[17])/1000.0:.2f}")

    # the slowest workgroups against the median: which phase makes the launch wait
    lifew = we - ws
    order = np.argsort(-lifew)[:6]
    cols = [("A", 0, 1), ("D1", 3, 4), ("D2", 4, 5), ("D3", 5, 6), ("P", 6, 8), ("E", 8, 7)]
    med = {nm: np.median((st[:, b] - st[:, a]) / 1000.0) for nm, a, b in cols}
    print("  median phases (kc):", " ".join(f"{nm} {med[nm]:.1f}" for nm, _, _ in cols), f"| life {np.median(lifew):.1f} us, items {np.median(tot):.0f}")
    for i in order:
        print(f"  slow wg: life {lifew[i]:.1f} us start {ws[i]:.2f} items {tot[i]:5d} |", " ".join(f"{nm} {(st[i, b] - st[i, a]) / 1000.0:.1f}" for nm, a, b in cols))
