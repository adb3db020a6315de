"""Diagnosis: marking update on a cloud wider than the window (set_cloud), device vs oracle, node by node."""
import os, sys
import numpy as np
from dddmr_navigation_amd import marking, scenes
from dddmr_navigation_amd.local_planner import LocalPlanner
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
T_BS = (0.0, 0.0, 0.5, 0, 0, 0, 1)
n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 16385
half = float(sys.argv[2]) if len(sys.argv) > 2 else 9.0
sc = scenes.bench_scene("C2")
cloud = sc.cloud
walls = cloud[(np.abs(np.abs(cloud[:, 1]) - 9.9) < 0.05)]
rng = np.random.default_rng(n_points)
near = cloud[(np.abs(cloud[:, 0] - 1.0) < half) & (np.abs(cloud[:, 1]) < half) & (cloud[:, 2] > 0.05) & (cloud[:, 2] < 2.0)]
cfg = marking.shipped_config(euclidean_cluster_extraction_tolerance=0.12)
ground = marking.ground_lattice()
print("ground lattice extent", ground.min(0), ground.max(0), len(ground), "obs extent", near[:, :3].min(0), near[:, :3].max(0))
mo = oracle.MarkingOracle(cfg, ground, walls[:, :3])
with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
    layer = marking.MarkingLayer(lp, cfg, ground, walls[:, :3])
    for k in range(3):
        pick = rng.choice(len(near), size=min(n_points, len(near)), replace=False)
        obs = np.ascontiguousarray(near[np.sort(pick)], dtype=np.float32)
        t_gb = (0.4 * k, 0.0, 0.0, 0, 0, 0, 1)
        lp.set_cloud(obs)
        st = layer.update(T_BS, t_gb)
        so = mo.update(obs[:, :3], T_BS, t_gb)
        dg, do = layer.dgraph(), mo.dgraph()
        bad = np.nonzero(dg != do)[0]
        print(f"update {k}: counts dev {(st.n_observation, st.n_clusters, st.n_marked, st.n_cleared, st.n_alive)} oracle {(so.n_observation, so.n_clusters, so.n_marked, so.n_cleared, so.n_alive)}; dgraph differs at {len(bad)} nodes, route {layer.route_counts()}")
        for i in bad[:12]:
            g = ground[i] if i < len(ground) else None
            d = np.hypot(obs[:, 0] - g[0], obs[:, 1] - g[1])
            near3 = np.linalg.norm(obs[:, :3] - g, axis=1)
            print(f"   node {i} at {g}: device {dg[i]:.6f} oracle {do[i]:.6f}; nearest raw point xy {d.min():.4f}, 3-D {near3.min():.4f}")
        (gp, gv), (op, ov) = layer.points(True), mo.points(True)
        from collections import defaultdict
        dg_, do_ = defaultdict(list), defaultdict(list)
        for q, v in zip(gp.tolist(), gv.tolist()): dg_[tuple(v)].append(tuple(q))
        for q, v in zip(op.tolist(), ov.tolist()): do_[tuple(v)].append(tuple(q))
        diff = [v for v in sorted(set(dg_) | set(do_)) if sorted(dg_.get(v, [])) != sorted(do_.get(v, []))]
        print(f"   generator points: device {len(gp)} in {len(dg_)} markings, oracle {len(op)} in {len(do_)}; markings whose point sets differ: {len(diff)}")
        for v in diff[:5]:
            a_, b_ = sorted(dg_.get(v, [])), sorted(do_.get(v, []))
            print(f"   voxel {v} (centre {v[0] * cfg.xy_resolution:.2f}, {v[1] * cfg.xy_resolution:.2f}, {v[2] * cfg.height_resolution:.2f}): device {len(a_)} points {a_[:4]} | oracle {len(b_)} points {b_[:4]}")
        if len(bad):
            break
