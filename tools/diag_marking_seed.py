import os, sys, math
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_marking_gpu as T
from dddmr_navigation_amd import marking, scenes
from dddmr_navigation_amd.local_planner import LocalPlanner
import oracle
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
_, _, walls, corridor = T._scene()
res = float(rng.choice([0.05, 0.1]))
cfg = marking.shipped_config(
    euclidean_cluster_extraction_tolerance=float(rng.choice([0.1, 0.15, 0.25])),
    euclidean_cluster_extraction_min_cluster_size=int(rng.choice([1, 3, 5])),
    segmentation_ignore_ratio=float(rng.choice([1.1, 0.3, 0.5, 0.7])),
    xy_resolution=res, height_resolution=res,
    inscribed_radius=float(rng.uniform(0.3, 0.6)), inflation_radius=float(rng.uniform(0.8, 1.6)),
    vertical_FOV_top=float(rng.choice([15.0, 20.0])), vertical_FOV_bottom=float(rng.choice([-15.0, -20.0])))
print("res", res, "tol", cfg.euclidean_cluster_extraction_tolerance, "min", cfg.euclidean_cluster_extraction_min_cluster_size)
static_map = walls if rng.random() < 0.5 else np.concatenate([walls, corridor])
n_updates = 8
xs = np.cumsum(rng.uniform(0.0, 0.4, n_updates)); ys = np.cumsum(rng.uniform(-0.15, 0.15, n_updates)); zs = np.cumsum(rng.uniform(-0.01, 0.02, n_updates))
rp = rng.uniform(-0.04, 0.04, (n_updates, 2))
holes = [(float(rng.uniform(0.0, 5.0)), float(rng.uniform(-2.0, 2.0)), float(rng.uniform(0.6, 1.6))) if rng.random() < 0.5 else None for _ in range(n_updates)]
sc, cloud, _, _ = T._scene()
ground = marking.ground_lattice()
mo = oracle.MarkingOracle(cfg, ground, static_map[:, :3])
with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
    layer = marking.MarkingLayer(lp, cfg, ground, static_map[:, :3])
    for k in range(n_updates):
        t_gb = (float(xs[k]), float(ys[k]), float(zs[k])) + tuple(scenes.quat_from_rpy(float(rp[k, 0]), float(rp[k, 1]), 0.0))
        c = cloud if holes[k] is None else cloud[np.hypot(cloud[:, 0] - holes[k][0], cloud[:, 1] - holes[k][1]) > holes[k][2]]
        scan = scenes.lidar_scan(c, sensor_xyz=(t_gb[0], t_gb[1], t_gb[2] + 0.5), seed=100 + k)
        lp.set_scan(scan, T.T_BS, t_gb, 5.0, 2.0)
        obs = lp.get_cloud()
        st = layer.update(T.T_BS, t_gb); so = mo.update(obs[:, :3], T.T_BS, t_gb)
        mv, mm, mf = mo.decisions(1)
        added = [tuple(v) for v, f in zip(mv.tolist(), mf) if f]
        from collections import Counter
        dup = [v for v, n in Counter(added).items() if n > 1]
        d = layer.dgraph(); o = mo.dgraph()
        bad = np.nonzero(d != o)[0]
        print("update", k, "added", len(added), "duplicate voxels among added", len(dup), dup[:4], "dgraph mismatches", len(bad), [(int(i), float(d[i]), float(o[i])) for i in bad[:4]])
