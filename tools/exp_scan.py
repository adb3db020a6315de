"""set_scan micro-benchmark: python tools/exp_scan.py"""
import time, numpy as np
from dddmr_navigation_amd import scenes, configs
from dddmr_navigation_amd.local_planner import LocalPlanner
sc = scenes.bench_scene("C2")
scans = [scenes.lidar_scan(sc.cloud, seed=100 + i) for i in range(10)]
tbs, tgb = (0.0, 0.0, 0.5, 0, 0, 0, 1), (0.0, 0.0, 0.0, 0, 0, 0, 1)
with LocalPlanner([sc.theory], max_points=len(sc.cloud)) as lp:
    lp.setPlan(sc.plan)
    for s in scans: lp.set_scan(s, tbs, tgb, 10.0, 2.0)
    t0 = time.perf_counter(); n = 0
    for r in range(20):
        for s in scans:
            n_out = lp.set_scan(s, tbs, tgb, 10.0, 2.0); n += 1
    dt = (time.perf_counter() - t0) / n
    print(f"set_scan: {len(scans[0])} pts -> {n_out} voxels, {dt*1e6:.1f} us per call")
    name = sc.theory.name.decode()
    t0 = time.perf_counter()
    for r in range(20):
        for s in scans:
            lp.set_scan(s, tbs, tgb, 10.0, 2.0); lp.tick(name, sc.tick)
    print(f"set_scan + tick: {(time.perf_counter()-t0)/200*1e6:.1f} us per step")
