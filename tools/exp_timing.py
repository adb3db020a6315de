"""Timing experiments on the GPU: python tools/exp_timing.py [C2|C3] [label]"""
import os, sys, time
import numpy as np
from dddmr_navigation_amd import scenes, _capi as K
from dddmr_navigation_amd.local_planner import LocalPlanner

def run(cfg, cloud_mode="full", iters=50):
    sc = scenes.bench_scene(cfg)
    if os.environ.get('EXP_NX'):
        sc.theory.linear_x_sample = float(os.environ['EXP_NX'])
    cloud = sc.cloud
    if cloud_mode == "empty":
        cloud = cloud[:0]
    elif cloud_mode == "far":
        cloud = cloud.copy(); cloud[:, 0] += 1000.0
    with LocalPlanner([sc.theory], max_points=max(len(cloud), 16)) as lp:
        lp.set_cloud(cloud); lp.setPlan(sc.plan)
        name = sc.theory.name.decode()
        for _ in range(5): lp.tick(name, sc.tick)
        dm, sm = [], []
        t0 = time.perf_counter()
        for _ in range(iters):
            r = lp.tick(name, sc.tick); dm.append(r.device_ms); sm.append(r.score_ms)
        wall = (time.perf_counter() - t0) / iters * 1e3
        print(f"{cfg} cloud={cloud_mode:5s} CELL={os.environ.get('DDDMR_CELL','-')} TILE={os.environ.get('DDDMR_TILE','-')} "
              f"binned={r.n_points_binned} wall_ms={wall:.3f} device_ms={np.median(dm):.3f} score_ms={np.median(sm):.3f} "
              f"min_score={np.min(sm):.3f} best={r.best_index}", flush=True)

if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    for mode in (sys.argv[2:] or ["full", "far", "empty"]):
        run(cfg, mode)
