"""Dump the observations / poses of tests/test_marking_gpu.py::test_marking_far_from_the_map_origin (offset 0) and the
device's voxel sets per update, for analysis on the CPU."""
import math, os, sys
import numpy as np
from dddmr_navigation_amd import marking, scenes
from dddmr_navigation_amd.local_planner import LocalPlanner
off = np.array((1500.0, -800.0, 30.0))
T_BS = (0.0, 0.0, 0.5, 0, 0, 0, 1)
sc = scenes.bench_scene("C2"); cloud = sc.cloud
walls = cloud[(np.abs(np.abs(cloud[:, 1]) - 9.9) < 0.05)]
sh = lambda a: np.concatenate([(a[:, :3].astype(np.float64) + off).astype(np.float32), a[:, 3:]], axis=1)
cloud_s, walls_s = sh(cloud), sh(walls)
ground = (marking.ground_lattice().astype(np.float64) + off).astype(np.float32)
cfg = marking.shipped_config()
gone = np.hypot(cloud_s[:, 0] - (off[0] + 2.0), cloud_s[:, 1] - off[1]) > 1.2
out = {}
with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
    layer = marking.MarkingLayer(lp, cfg, ground, walls_s[:, :3])
    for k in range(9):
        t_gb = (off[0] + 0.25 * k, off[1] + 0.05 * k, off[2], 0, 0, math.sin(0.03 * k), math.cos(0.03 * k))
        scene = cloud_s if k < 5 else cloud_s[gone]
        scan = scenes.lidar_scan(scene, sensor_xyz=(t_gb[0], t_gb[1], t_gb[2] + 0.5), seed=100 + k)
        lp.set_scan(scan, T_BS, t_gb, 5.0, 2.0)
        out[f"obs{k}"] = lp.get_cloud()
        out[f"pose{k}"] = np.array(t_gb)
        st = layer.update(T_BS, t_gb)
        out[f"vox{k}"] = layer.voxels()
        out[f"cnt{k}"] = np.array([st.n_observation, st.n_clusters, st.n_marked, st.n_in_window, st.n_cleared, st.n_alive])
os.makedirs("gpurun_out/r03", exist_ok=True)
np.savez_compressed("gpurun_out/r03/far_dump.npz", **out)
print("saved", {k: v.shape for k, v in out.items() if k.startswith("obs")})
