"""PCIe-inclusive rate: the boundary hands over HOST buffers, so a tick that has to take a new
aggregate observation pays set_cloud (repack to 16-byte records + H2D) first.
usage: python tools/exp_pcie.py [C2|C3]"""
import sys, time
import numpy as np
from dddmr_navigation_amd import scenes
from dddmr_navigation_amd.local_planner import LocalPlanner

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
sc = scenes.bench_scene(cfg)
name = sc.theory.name.decode()
xyzi32 = np.zeros((len(sc.cloud), 8), np.float32)      # pcl::PointXYZI records (stride 32)
xyzi32[:, :4] = sc.cloud
with LocalPlanner([sc.theory], max_points=len(sc.cloud)) as lp:
    lp.setPlan(sc.plan)
    for label, cloud in (("packed xyzi, stride 16", sc.cloud), ("pcl::PointXYZI, stride 32", xyzi32)):
        for _ in range(10):
            lp.set_cloud(cloud); r = lp.tick(name, sc.tick)
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            lp.set_cloud(cloud)
        t_set = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            lp.set_cloud(cloud); r = lp.tick(name, sc.tick)
        t_both = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            r = lp.tick(name, sc.tick)
        t_tick = (time.perf_counter() - t0) / n
        print(f"{cfg} {label}: set_cloud {t_set*1e6:.1f} us ({cloud.nbytes/t_set/1e9:.2f} GB/s of caller bytes), "
              f"tick {t_tick*1e6:.1f} us, set_cloud+tick {t_both*1e6:.1f} us "
              f"=> {r.n_samples/t_both/1e6:.1f} M trajectories/s PCIe-inclusive vs {r.n_samples/t_tick/1e6:.1f} M resident", flush=True)
