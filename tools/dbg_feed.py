"""Determinism probe: set_scan -> tick, set_cloud -> tick, tick again; prints how many per-trajectory
costs differ between the ticks (they must not: any tile assignment gives identical results) and
against the oracle.  This is how the backend-fused copy of the 1-NN loop was found.
usage: python tools/dbg_feed.py"""
import numpy as np, sys
sys.path.insert(0,'.')
from dddmr_navigation_amd import scenes
from dddmr_navigation_amd.local_planner import LocalPlanner
import oracle
sc = scenes.bench_scene("C2")
scan = scenes.lidar_scan(sc.cloud, seed=9)
tbs = (0.0, 0.0, 0.5, 0, 0, 0, 1); tgb = (0.0, 0.0, 0.0, 0, 0, 0, 1)
ref = oracle.feed(scan, tbs, tgb, 10.0, 2.0)
cl = np.concatenate([ref, np.zeros((len(ref), 1), np.float32)], axis=1)
name = sc.theory.name.decode()
o = oracle.tick(sc.theory, cl, sc.plan, sc.tick, n_threads=8, want_margin=True)
fragile = np.abs(o.min_margin) < 1e-4
for rep in range(6):
    with LocalPlanner([sc.theory], max_points=40_000) as lp:
        lp.setPlan(sc.plan)
        lp.set_scan(scan, tbs, tgb, 10.0, 2.0)
        r1 = lp.tick(name, sc.tick); c1 = lp.debug()[0].copy(); s1 = lp.debug()[1].copy()
        lp.set_cloud(cl)
        r2 = lp.tick(name, sc.tick); c2 = lp.debug()[0].copy()
        r3 = lp.tick(name, sc.tick); c3 = lp.debug()[0].copy()
    for nm, c in (("c1", c1), ("c2", c2), ("c3", c3)):
        bad = np.nonzero((c != o.costs) & ~fragile & (np.abs(c - o.costs) > 1e-9))[0]
    d = np.nonzero(c2 != c3)[0]
    print("rep", rep, "c1!=c2", int(((c1 != c2) & ~fragile).sum()), "c2!=c3", len(d), [(int(i), float(c2[i]), float(c3[i] - c2[i])) for i in d[:4]], flush=True)
