import time, numpy as np
from dddmr_navigation_amd import scenes
from dddmr_navigation_amd.local_planner import LocalPlanner
sc = scenes.bench_scene("C3")
with LocalPlanner([sc.theory], max_points=len(sc.cloud), max_trajectories=1 << 15) as lp:
    lp.set_cloud(sc.cloud); lp.setPlan(sc.plan)
    name = sc.theory.name.decode()
    ts = []
    for i in range(300):
        t0 = time.perf_counter(); lp.tick(name, sc.tick); ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.array(ts)
    print("tick us:", " ".join(f"{v:.0f}" for v in ts[:40]))
    for a, b in ((0, 5), (5, 25), (25, 50), (50, 100), (100, 200), (200, 300)):
        print(f"ticks {a}-{b}: median {np.median(ts[a:b]):.1f} mean {ts[a:b].mean():.1f}")
    time.sleep(0.5)
    ts2 = []
    for i in range(40):
        t0 = time.perf_counter(); lp.tick(name, sc.tick); ts2.append((time.perf_counter() - t0) * 1e6)
    print("after 0.5 s idle:", " ".join(f"{v:.0f}" for v in ts2[:25]))
