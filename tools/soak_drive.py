"""Stability soak of the whole chain a deployment runs every control tick: lidar scan -> set_scan (voxel feed,
stitcher) -> marking / clearing layer update -> tick, with the robot driving laps in the C2 scene for N steps
(no oracle: it checks that nothing errors, hangs, leaks or drifts -- alive markings stay bounded, every tick returns a
command or a clean ALL_TRAJECTORIES_FAIL).  usage: python tools/soak_drive.py [steps=3000] [overlapped]
(overlapped: tick_begin -> marking update -> tick_end, the update on its own stream next to the tick's kernels)"""
import math, sys, time
import numpy as np
from dddmr_navigation_amd import _capi as K, configs, marking, scenes
from dddmr_navigation_amd.local_planner import LocalPlanner

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
overlapped = len(sys.argv) > 2 and sys.argv[2] == "overlapped"
sc = scenes.bench_scene("C2")
th = configs.omni_simple_shipped(linear_x_sample=8.0, linear_y_sample=8.0, angular_z_sample=8.0)
walls = sc.cloud[(np.abs(np.abs(sc.cloud[:, 1]) - 9.9) < 0.05)]
T_BS = (0.0, 0.0, 0.5, 0, 0, 0, 1)
rng = np.random.default_rng(1)
states = {}
alive, marked, cleared, t_upd, t_tick = [], 0, 0, [], []
t0 = time.time()
with LocalPlanner([th], max_points=1 << 16) as lp:
    layer = marking.MarkingLayer(lp, marking.shipped_config(max_markings=1 << 18, max_cluster_points=1 << 22), marking.ground_lattice(), walls[:, :3])
    lp.set_stitcher(2)
    lp.setPlan(sc.plan)
    for k in range(n_steps):
        a = 0.004 * k                                  # laps of an ellipse inside the corridor
        x, y, yaw = 6.0 * math.sin(a), 2.5 * math.sin(2 * a), math.atan2(5.0 * math.cos(2 * a), 6.0 * math.cos(a))
        t_gb = (x, y, 0.0) + tuple(scenes.quat_from_rpy(0.02 * math.sin(0.1 * k), 0.02 * math.cos(0.07 * k), yaw))
        cloud = sc.cloud if (k // 40) % 3 else sc.cloud[np.hypot(sc.cloud[:, 0] - x - 1.5 * math.cos(yaw), sc.cloud[:, 1] - y - 1.5 * math.sin(yaw)) > 1.0]
        scan = scenes.lidar_scan(cloud, sensor_xyz=(x, y, 0.5), seed=int(rng.integers(1 << 20)))
        lp.set_scan(scan, T_BS, t_gb, 5.0, 2.0)
        tick_in = scenes.tick_input(pose=t_gb, twist=(0.4, 0.0, 0.1 * math.sin(0.05 * k)))
        t1 = time.perf_counter()
        if overlapped:
            lp.tick_begin(th.name.decode(), tick_in)
            st = layer.update(T_BS, t_gb)
            t2 = time.perf_counter()
            res = lp.tick_end()
        else:
            st = layer.update(T_BS, t_gb)
            t2 = time.perf_counter()
            res = lp.tick(th.name.decode(), tick_in)
        t3 = time.perf_counter()
        assert res.planner_state in (K.TRAJECTORY_FOUND, K.ALL_TRAJECTORIES_FAIL), res.planner_state
        assert res.best_index >= 0 or res.planner_state == K.ALL_TRAJECTORIES_FAIL
        states[res.planner_state] = states.get(res.planner_state, 0) + 1
        alive.append(st.n_alive); marked += st.n_marked; cleared += st.n_cleared
        t_upd.append(t2 - t1); t_tick.append(t3 - t2)
        if k % 500 == 499:
            print(f"step {k + 1}: alive {st.n_alive}, marked so far {marked}, cleared {cleared}, update {1e3 * np.median(t_upd[-500:]):.3f} ms, "
                  f"tick {1e3 * np.median(t_tick[-500:]):.3f} ms, states {states}", flush=True)
print(f"{n_steps} steps in {time.time() - t0:.1f} s: alive min {min(alive)} max {max(alive)}, marked {marked}, cleared {cleared}, "
      f"update median {1e3 * np.median(t_upd):.3f} ms (max {1e3 * max(t_upd):.2f}), tick median {1e3 * np.median(t_tick):.3f} ms (max {1e3 * max(t_tick):.2f}), states {states}")
