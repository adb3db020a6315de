"""Throughput with several independent contexts (robots) sharing one GPU: M ticks in flight,
each context ticking sequentially.  usage: python tools/exp_contexts.py [C2] [contexts...]"""
import sys, time
import numpy as np
from dddmr_navigation_amd import scenes
from dddmr_navigation_amd.local_planner import LocalPlanner

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
counts = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
sc = scenes.bench_scene(cfg)
name = sc.theory.name.decode()
for m in counts:
    lps = [LocalPlanner([sc.theory], max_points=len(sc.cloud), max_trajectories=1 << 17) for _ in range(m)]
    for lp in lps:
        lp.set_cloud(sc.cloud); lp.setPlan(sc.plan)
        for _ in range(10):
            lp.tick(name, sc.tick)
    n = 400
    # every context keeps exactly one tick in flight: begin on all, then end/begin round robin
    for lp in lps:
        lp.tick_begin(name, sc.tick)
    t0 = time.perf_counter()
    done = 0
    i = 0
    while done < n:
        lp = lps[i % m]
        r = lp.tick_end()
        done += 1
        lp.tick_begin(name, sc.tick)
        i += 1
    el = time.perf_counter() - t0
    for lp in lps:
        r = lp.tick_end()
    print(f"{cfg} contexts={m}: {el / n * 1e6:.1f} us per tick, {r.n_samples * n / el / 1e6:.1f} M trajectories/s, best={r.best_index}", flush=True)
    for lp in lps:
        lp.close()
