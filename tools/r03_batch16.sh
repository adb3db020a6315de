#!/bin/bash
mkdir -p gpurun_out/r03
export PYTHONPATH=$PWD
python bench.py --workload C5M --steps 300 --warmup 50 --no-ceiling --no-cpu-baseline > gpurun_out/r03/c5m_b16.json 2> gpurun_out/r03/c5m_b16.err
python -c "import json; d=json.load(open('gpurun_out/r03/c5m_b16.json')); m=d['config']['marking']; print('C5M', d['ms_per_step'], m['serial_schedule_ms_per_step'], m['clear_ms'], m['mark_ms'], d['config']['cmd_vel_matches_oracle'])" || tail -5 gpurun_out/r03/c5m_b16.err
python - <<'PY'
# update time of a 27000-point observation (a 32-line lidar's worth) on both routes
import os, time, numpy as np
from dddmr_navigation_amd import marking, scenes
from dddmr_navigation_amd.local_planner import LocalPlanner
sc = scenes.bench_scene("C2"); cloud = sc.cloud
walls = cloud[(np.abs(np.abs(cloud[:, 1]) - 9.9) < 0.05)]
near = cloud[(np.abs(cloud[:, 0] - 1.0) < 9.0) & (np.abs(cloud[:, 1]) < 9.0) & (cloud[:, 2] > 0.05) & (cloud[:, 2] < 2.0)]
rng = np.random.default_rng(7)
obs = [np.ascontiguousarray(near[np.sort(rng.choice(len(near), size=27000, replace=False))], dtype=np.float32) for _ in range(6)]
cfg = marking.shipped_config(euclidean_cluster_extraction_tolerance=0.075, perception_window_size=10.0)
for route in ("fused", "general"):
    os.environ["DDDMR_MARKING_ROUTE"] = route
    with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
        layer = marking.MarkingLayer(lp, cfg, marking.ground_lattice(), walls[:, :3])
        ts = []
        for k in range(40):
            lp.set_cloud(obs[k % 6])
            t_gb = (0.05 * (k % 20), 0.0, 0.0, 0, 0, 0, 1)
            t0 = time.perf_counter(); st = layer.update((0, 0, 0.5, 0, 0, 0, 1), t_gb); ts.append(time.perf_counter() - t0)
        print(route, "27000 points: update median ms", round(1e3 * float(np.median(ts[10:])), 4), "alive", st.n_alive, "clusters", st.n_clusters, layer.route_counts())
PY
