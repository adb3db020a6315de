"""Long horizons (rotate-in-place with a fine angular granularity): parity at ~600 steps, and the
clean capacity error beyond what one workgroup's LDS holds.  usage: python tools/exp_long_horizon.py"""
import numpy as np, sys
from dddmr_navigation_amd import configs, scenes, _capi as K
from dddmr_navigation_amd.local_planner import LocalPlanner, RolloutError
import oracle
sc = scenes.bench_scene("C1")
post = np.array([[0.55, 0.1, 0.3, 0]] * 8, np.float32)
cloud = np.concatenate([sc.cloud, post])
for gran, ms in ((0.0105, 1024), (0.004, 2048), (0.002, 4096)):
    th = configs.rotate_inplace_shipped("rot", angular_sim_granularity=gran)
    try:
        with LocalPlanner([th], max_points=len(cloud), max_steps=ms) as lp:
            lp.set_cloud(cloud); lp.setPlan(sc.plan)
            r = lp.tick("rot", scenes.tick_input())
            c, s, smp = lp.debug()
        o = oracle.tick(th, cloud, sc.plan, scenes.tick_input())
        print(gran, "steps", s, "costs", c, "oracle", o.costs, o.steps, "best", r.best_index, o.result.best_index)
    except RolloutError as e:
        print(gran, "error", e)
