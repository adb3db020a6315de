"""Global-mode marking / clearing layer: the HIP path through the C-ABI
(dddmr_rollout_marking_*) against the CPU restatement (oracle/oracle_marking.cpp) of
MultiLayerSpinningLidar::selfClear / selfMark + Marking::addPCPtr / removePCPtr, update by update over
scan sequences of the C2 scene: the set of stored voxels, the dGraph, the lethal set and the per-update
counts must be IDENTICAL (integer / set work bit-exact; dGraph distances are floats formed by the same
operations in the same order, so they are compared exactly too).

Both sides get the same observation (the cloud the device feed produced, in its order): the summation
order of a cluster's centroid is the point order of pcl_msg_gbl_, an upstream artefact."""
import math
import os

import numpy as np
import pytest

from dddmr_navigation_amd import _capi as K, marking, scenes
from dddmr_navigation_amd.local_planner import LocalPlanner, RolloutError
import oracle

pytestmark = pytest.mark.gpu
T_BS = (0.0, 0.0, 0.5, 0, 0, 0, 1)


def _scene():
    sc = scenes.bench_scene("C2")
    cloud = sc.cloud
    walls = cloud[(np.abs(np.abs(cloud[:, 1]) - 9.9) < 0.05)]
    corridor = cloud[(np.abs(np.abs(cloud[:, 1]) - 4.5) < 0.05)]
    return sc, cloud, walls, corridor


def _vset(v):
    return set(map(tuple, np.asarray(v).tolist()))


@pytest.fixture(autouse=True, params=["fused", "general"])
def route(request, monkeypatch):
    """Every test runs on both routes of dddmr_rollout_marking_update: `fused` (five launches, observations of up to
    16384 points: csrc/marking_fused.hip.h) and `general` (library sorts, any size: csrc/marking.hip.h).  The layer
    reads DDDMR_MARKING_ROUTE when it is created."""
    monkeypatch.setenv("DDDMR_MARKING_ROUTE", request.param)
    return request.param


def _check_route(layer, n_updates):
    want = os.environ.get("DDDMR_MARKING_ROUTE")
    rc = layer.route_counts()
    if want == "fused":
        assert rc["fused"] == n_updates and rc["general"] == 0, rc
        assert rc["launches_last_update"] <= 12, rc          # <= 12 launches per update
    elif want == "general":
        assert rc["general"] == n_updates and rc["fused"] == 0, rc


STATS = {"sequences": 0, "updates_compared": 0, "sequences_stopped_at_a_fragile_decision": 0, "smallest_margin_of_a_stop": None}


def _run_sequence(cfg, static_map, poses, scene_of, window=5.0, height=2.0, ground=None, n_updates=10, fragile_tol=None):
    """fragile_tol: a selfClear / selfMark decision whose oracle margin (distance of the deciding quantity from its
    threshold: FOV angles in degrees, ray distances and voxel borders in metres) is below it may legitimately fall the
    other way on the device (asin / atan2 of ocml vs glibc differ in the last place); the two stores then differ from
    that update on, so the sequence stops there and is counted.  None = every difference fails."""
    sc, cloud, _, _ = _scene()
    ground = marking.ground_lattice() if ground is None else ground
    mo = oracle.MarkingOracle(cfg, ground, static_map[:, :3])
    totals = dict(marked=0, cleared=0, clusters=0)
    overlap = os.environ.get("DDDMR_MARKING_OVERLAP", "0") not in ("", "0")   # every update next to a pending tick
    with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
        layer = marking.MarkingLayer(lp, cfg, ground, static_map[:, :3])
        if overlap:
            lp.setPlan(sc.plan)
        for k in range(n_updates):
            t_gb = poses(k)
            scan = scenes.lidar_scan(scene_of(k, cloud), sensor_xyz=(t_gb[0], t_gb[1], t_gb[2] + 0.5), seed=100 + k)
            lp.set_scan(scan, T_BS, t_gb, window, height)
            obs = lp.get_cloud()
            if overlap:
                lp.tick_begin(sc.theory.name.decode(), scenes.tick_input(pose=t_gb))
                st = layer.update(T_BS, t_gb)
                assert lp.tick_end().planner_state in (K.TRAJECTORY_FOUND, K.ALL_TRAJECTORIES_FAIL)
            else:
                st = layer.update(T_BS, t_gb)
            so = mo.update(obs[:, :3], T_BS, t_gb)
            got = (st.n_observation, st.n_clusters, st.n_marked, st.n_in_window, st.n_cleared, st.n_alive)
            want = (so.n_observation, so.n_clusters, so.n_marked, so.n_in_window, so.n_cleared, so.n_alive)
            gv, ov = _vset(layer.voxels()), _vset(mo.voxels())
            if gv != ov or got != want:
                cv, cm, cf = mo.decisions(0)
                mv, mm, mf = mo.decisions(1)
                marg = {tuple(v): float(m) for v, m in zip(cv.tolist(), cm)}
                marg.update({tuple(v): min(float(m), marg.get(tuple(v), 1e9)) for v, m in zip(mv.tolist(), mm)})
                diff = sorted(gv ^ ov)
                if fragile_tol is not None and diff and all(marg.get(d, 1e9) < fragile_tol for d in diff):
                    STATS["sequences_stopped_at_a_fragile_decision"] += 1
                    mn = min(marg[d] for d in diff)
                    STATS["smallest_margin_of_a_stop"] = mn if STATS["smallest_margin_of_a_stop"] is None else min(mn, STATS["smallest_margin_of_a_stop"])
                    STATS["sequences"] += 1
                    return totals, None
                pytest.fail(f"update {k}: counts {got} vs oracle {want}; {len(diff)} voxels differ, "
                            f"oracle margins of the first: {[(d, marg.get(d)) for d in diff[:6]]}")
            np.testing.assert_array_equal(layer.lethal(), mo.lethal())
            np.testing.assert_array_equal(layer.dgraph(), mo.dgraph())
            totals["marked"] += so.n_marked; totals["cleared"] += so.n_cleared; totals["clusters"] += so.n_clusters
            STATS["updates_compared"] += 1
        final = (len(gv), int((mo.dgraph() < cfg.max_obstacle_distance).sum()), int(mo.lethal().sum()))
        _check_route(layer, n_updates)
        layer.reset()
        assert len(layer.voxels()) == 0 and (layer.dgraph() == cfg.max_obstacle_distance).all() and not layer.lethal().any()
    STATS["sequences"] += 1
    return totals, final


def test_shipped_config_ten_scans_with_a_vanishing_obstacle():
    """The shipped global lidar block (tolerance 0.1, resolution 0.05, static check off): the robot drives
    3 m down the corridor; from scan 5 on everything within 1.5 m of (2.5, 0) is gone, so the rays that used
    to stop there pass and selfClear removes those markings."""
    _, _, walls, _ = _scene()
    cfg = marking.shipped_config()
    poses = lambda k: (0.3 * k, 0.0, 0.0, 0, 0, 0, 1)
    scene_of = lambda k, cloud: cloud if k < 5 else cloud[np.hypot(cloud[:, 0] - 2.5, cloud[:, 1]) > 1.5]
    totals, final = _run_sequence(cfg, walls, poses, scene_of)
    assert totals["marked"] > 5000 and totals["cleared"] > 3000 and final[0] > 1000 and final[2] > 100


def test_pool_compaction_keeps_the_store_identical():
    """A pool of generator points barely larger than what the alive markings need: cleared and replaced markings
    leave garbage behind, so the compaction pass (alive markings move to the front of the second pool buffer) runs
    every few updates -- results must not move."""
    _, _, walls, _ = _scene()
    cfg = marking.shipped_config(max_cluster_points=24000)
    poses = lambda k: (0.3 * k, 0.0, 0.0, 0, 0, 0, 1)
    scene_of = lambda k, cloud: cloud if k < 5 else cloud[np.hypot(cloud[:, 0] - 2.5, cloud[:, 1]) > 1.5]
    totals, final = _run_sequence(cfg, walls, poses, scene_of)
    assert totals["marked"] > 5000 and final[0] > 1000


def test_store_garbage_collection_keeps_the_store_identical():
    """A table of 16384 slots for ~8000 alive markings: voxels whose marking was cleared keep their key, so the table
    passes half full within a few updates of a moving robot and the garbage collection (alive markings move to a fresh
    table, dead keys are dropped: the reference's entries with has_pc false, which nothing reads) runs several times
    -- results must not move, and the layer must not report a full table."""
    _, _, walls, _ = _scene()
    cfg = marking.shipped_config(max_markings=1 << 14)
    poses = lambda k: (0.35 * k, 0.0, 0.0, 0, 0, 0, 1)
    scene_of = lambda k, cloud: cloud if k % 3 else cloud[np.hypot(cloud[:, 0] - 0.35 * k - 1.5, cloud[:, 1]) > 1.3]
    totals, final = _run_sequence(cfg, walls, poses, scene_of, n_updates=16)
    assert totals["marked"] > 16384 and totals["cleared"] > 3000


def test_coarse_clusters_static_map_and_tilted_robot():
    """Tolerance 0.25 / min cluster size 3 (large wall clusters -> long summation chains), static-map
    rejection on (segmentation_ignore_ratio 0.5, the corridor walls are the static map), robot pitched and
    rolled by a few degrees and yawing (the projection plane and the FOV test follow base_link)."""
    _, _, walls, corridor = _scene()
    cfg = marking.shipped_config(euclidean_cluster_extraction_tolerance=0.25, euclidean_cluster_extraction_min_cluster_size=3,
                                 segmentation_ignore_ratio=0.5, xy_resolution=0.1, height_resolution=0.1,
                                 inscribed_radius=0.35, inflation_radius=1.0)
    static_map = np.concatenate([walls, corridor])

    def poses(k):
        q = scenes.quat_from_rpy(0.03 * math.sin(k), 0.04 * math.cos(k), 0.0)     # (the synthetic scan keeps global axes: yaw 0)
        return (0.25 * k, 0.1 * math.sin(k), 0.02 * k) + q
    scene_of = lambda k, cloud: cloud if k % 4 else cloud[np.hypot(cloud[:, 0] - 3.0, cloud[:, 1] + 1.0) > 1.2]
    totals, final = _run_sequence(cfg, static_map, poses, scene_of, n_updates=8)
    assert totals["marked"] > 100 and totals["cleared"] > 10


def test_small_observation_is_not_marked_and_counts_as_clear():
    """<= 5 points: selfMark returns before touching pcl_msg_gbl_ (:320-321)."""
    _, cloud, walls, _ = _scene()
    cfg = marking.shipped_config()
    ground = marking.ground_lattice()
    mo = oracle.MarkingOracle(cfg, ground, walls[:, :3])
    sc = scenes.bench_scene("C2")
    with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
        layer = marking.MarkingLayer(lp, cfg, ground, walls[:, :3])
        t_gb = (0.0, 0.0, 0.0, 0, 0, 0, 1)
        tiny = np.array([[2.0, 0.1 * i, 0.5, 0] for i in range(4)], dtype=np.float32)
        full = scenes.lidar_scan(cloud, sensor_xyz=(0, 0, 0.5), seed=5)
        for obs_scan in (tiny, full, tiny, full):
            lp.set_scan(obs_scan, T_BS, t_gb, 5.0, 2.0)
            obs = lp.get_cloud()
            st = layer.update(T_BS, t_gb)
            so = mo.update(obs[:, :3], T_BS, t_gb)
            assert (st.n_observation, st.n_clusters, st.n_marked, st.n_in_window, st.n_cleared, st.n_alive) == \
                   (so.n_observation, so.n_clusters, so.n_marked, so.n_in_window, so.n_cleared, so.n_alive)
            assert _vset(layer.voxels()) == _vset(mo.voxels())
            np.testing.assert_array_equal(layer.dgraph(), mo.dgraph())


def test_marking_capacity_and_state_errors():
    sc = scenes.bench_scene("C1")
    with LocalPlanner([sc.theory], max_points=1 << 14) as lp:
        with pytest.raises(RolloutError) as e:
            lp._check(lp._lib.dddmr_rollout_marking_update(lp._ctx, (7 * K.C.c_double)(), (7 * K.C.c_double)(), None))
        assert e.value.code == K.ERR_STATE
        _, cloud, walls, _ = _scene()
        layer = marking.MarkingLayer(lp, marking.shipped_config(max_markings=64), marking.ground_lattice(), walls[:, :3])
        lp.set_scan(scenes.lidar_scan(cloud, sensor_xyz=(0, 0, 0.5), seed=1), T_BS, (0, 0, 0, 0, 0, 0, 1), 5.0, 2.0)
        with pytest.raises(RolloutError) as e:
            layer.update(T_BS, (0, 0, 0, 0, 0, 0, 1))
        assert e.value.code == K.ERR_CAPACITY


# DDDMR_MARKING_SEEDS=N widens the sweep for a soak run (default 3 keeps the suite short)


@pytest.mark.parametrize("seed", range(int(os.environ.get("DDDMR_MARKING_SEEDS", "3"))))
def test_random_marking_sequences(seed):
    """Randomised layer parameters, robot paths (position, small roll / pitch, height drift) and obstacles that come
    and go: every update's voxel set, counts, dGraph and lethal set against the oracle, as in the fixed sequences."""
    rng = np.random.default_rng(1000 + seed + int(os.environ.get("DDDMR_SEED_BASE", "0")))
    _, _, walls, corridor = _scene()
    res = float(rng.choice([0.05, 0.1]))
    cfg = marking.shipped_config(
        euclidean_cluster_extraction_tolerance=float(rng.choice([0.1, 0.15, 0.25])),
        euclidean_cluster_extraction_min_cluster_size=int(rng.choice([1, 3, 5])),
        segmentation_ignore_ratio=float(rng.choice([1.1, 0.3, 0.5, 0.7])),
        xy_resolution=res, height_resolution=res,
        inscribed_radius=float(rng.uniform(0.3, 0.6)), inflation_radius=float(rng.uniform(0.8, 1.6)),
        vertical_FOV_top=float(rng.choice([15.0, 20.0])), vertical_FOV_bottom=float(rng.choice([-15.0, -20.0])))
    tilt = 0.04
    if os.environ.get("DDDMR_RANDOM_WILD", "0") not in ("", "0"):
        # soak runs: the layer's other parameters move too (own stream, seeded without drawing from rng)
        wr = np.random.default_rng(int(rng.bit_generator.state["state"]["state"]) & 0xFFFFFFFF)
        cfg.height_resolution = float(wr.choice([0.05, 0.1, 0.2]))
        cfg.marking_height = float(wr.uniform(0.8, 2.5))
        cfg.perception_window_size = float(wr.uniform(3.0, 8.0))
        a0, a1 = float(wr.uniform(5.0, 60.0)), float(wr.uniform(120.0, 180.0))
        cfg.scan_effective_positive_start, cfg.scan_effective_positive_end = a0, a1
        cfg.scan_effective_negative_start, cfg.scan_effective_negative_end = -float(wr.uniform(5.0, 60.0)), -float(wr.uniform(120.0, 180.0))
        cfg.vertical_FOV_top, cfg.vertical_FOV_bottom = float(wr.uniform(5.0, 30.0)), -float(wr.uniform(5.0, 30.0))
        tilt = float(wr.choice([0.04, 0.15, 0.3]))
        yaw_step = float(wr.choice([0.0, 0.2, 0.9])) if os.environ.get("DDDMR_RANDOM_YAW", "0") not in ("", "0") else 0.0
        yaw0 = float(wr.uniform(-math.pi, math.pi)) if yaw_step else 0.0
    else:
        yaw_step, yaw0 = 0.0, 0.0
    static_map = walls if rng.random() < 0.5 else np.concatenate([walls, corridor])
    n_updates = 8
    xs = np.cumsum(rng.uniform(0.0, 0.4, n_updates))
    ys = np.cumsum(rng.uniform(-0.15, 0.15, n_updates))
    zs = np.cumsum(rng.uniform(-0.01, 0.02, n_updates))
    rp = rng.uniform(-0.04, 0.04, (n_updates, 2)) * (tilt / 0.04)
    holes = [(float(rng.uniform(0.0, 5.0)), float(rng.uniform(-2.0, 2.0)), float(rng.uniform(0.6, 1.6)))
             if rng.random() < 0.5 else None for _ in range(n_updates)]

    def poses(k):
        # (DDDMR_RANDOM_YAW with DDDMR_RANDOM_WILD: the robot also turns, from any heading -- the azimuth limits of
        # isinLidarObservation and the wrap of its shortest-angle at +-pi)
        return (float(xs[k]), float(ys[k]), float(zs[k])) + tuple(scenes.quat_from_rpy(float(rp[k, 0]), float(rp[k, 1]), yaw0 + yaw_step * k))

    def scene_of(k, cloud):
        if holes[k] is None:
            return cloud
        hx, hy, hr = holes[k]
        return cloud[np.hypot(cloud[:, 0] - hx, cloud[:, 1] - hy) > hr]

    ground = None
    if os.environ.get("DDDMR_RANDOM_SHIFT"):
        # the whole scenario moved by a map-scale offset ("x,y,z" in metres)
        off = np.array([float(v) for v in os.environ["DDDMR_RANDOM_SHIFT"].split(",")], dtype=np.float64)
        _, cloud0, _, _ = _scene()
        sh = lambda a: np.concatenate([(a[:, :3].astype(np.float64) + off).astype(np.float32), a[:, 3:]], axis=1)
        cloud_s, static_map = sh(cloud0), sh(static_map)
        ground = (marking.ground_lattice().astype(np.float64) + off).astype(np.float32)
        poses0, scene0 = poses, scene_of
        poses = lambda k: tuple(np.array(poses0(k)[:3]) + off) + tuple(poses0(k)[3:])
        scene_of = lambda k, c: sh(scene0(k, cloud0))
    totals, final = _run_sequence(cfg, static_map, poses, scene_of, ground=ground, n_updates=n_updates, fragile_tol=1e-5)
    assert totals["clusters"] > 0 or final is None


def teardown_module(module):
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_stats_marking.json"), "w") as f:
        json.dump(STATS, f, indent=1)
    print("\n[parity stats, marking layer]", STATS)


def test_long_sequence_with_many_compactions():
    """A long drive up and down the corridor with obstacles that come and go and a pool barely larger than what the
    alive markings need (compaction every few updates): the store must stay identical to the oracle's all the way
    (DDDMR_MARKING_LONG=N updates, default 24)."""
    _, _, walls, _ = _scene()
    n = int(os.environ.get("DDDMR_MARKING_LONG", "24"))
    cfg = marking.shipped_config(max_cluster_points=150000)
    poses = lambda k: (3.0 * math.sin(0.11 * k), 0.4 * math.sin(0.05 * k), 0.0) + tuple(scenes.quat_from_rpy(0.0, 0.0, 0.0))

    def scene_of(k, cloud):
        cx = 2.5 * math.sin(0.07 * k + 1.0)
        return cloud if (k // 5) % 2 == 0 else cloud[np.hypot(cloud[:, 0] - cx, cloud[:, 1]) > 1.2]

    totals, final = _run_sequence(cfg, walls, poses, scene_of, n_updates=n, fragile_tol=1e-5)
    assert totals["marked"] > 1000 and totals["cleared"] > 100


@pytest.mark.parametrize("n_points,tol", [(3000, 0.12), (4096, 0.12), (4097, 0.12), (8192, 0.12), (8193, 0.12), (15000, 0.12), (16384, 0.12),
                                          (16385, 0.075), (20481, 0.075), (27000, 0.075), (32768, 0.075), (27000, 0.12)])
def test_observation_sizes_of_every_fused_instantiation(n_points, tol, route):
    """Clouds handed over with set_cloud at and around the sizes where the fused route changes gear (the grid builder keeps
    16 points per lane in registers up to 16384 points and parks the rest in global memory up to 32768, the seed scan
    and the radix sorts change their batch counts at multiples of 4096), three updates each with the robot moving.  The
    bigger clouds reach 4 m beyond the layer's window on every side (a cloud handed over uncropped; the general route used
    to drop points more than 3.2 m outside it without a word), and with the wide tolerance their largest cluster (5077
    points) exceeds a partition workgroup: the fused route must hand the mark phase to the general one."""
    _, cloud, walls, _ = _scene()
    rng = np.random.default_rng(n_points)
    half = 5.0 if n_points <= 16384 else 9.0
    near = cloud[(np.abs(cloud[:, 0] - 1.0) < half) & (np.abs(cloud[:, 1]) < half) & (cloud[:, 2] > 0.05) & (cloud[:, 2] < 2.0)]
    cfg = marking.shipped_config(euclidean_cluster_extraction_tolerance=tol)
    ground = marking.ground_lattice()
    mo = oracle.MarkingOracle(cfg, ground, walls[:, :3])
    sc = scenes.bench_scene("C2")
    with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
        layer = marking.MarkingLayer(lp, cfg, ground, walls[:, :3])
        for k in range(3):
            pick = rng.choice(len(near), size=min(n_points, len(near)), replace=False)
            obs = np.ascontiguousarray(near[np.sort(pick)], dtype=np.float32)
            assert len(obs) == n_points
            t_gb = (0.4 * k, 0.0, 0.0, 0, 0, 0, 1)
            lp.set_cloud(obs)
            st = layer.update(T_BS, t_gb)
            so = mo.update(obs[:, :3], T_BS, t_gb)
            assert (st.n_observation, st.n_clusters, st.n_marked, st.n_in_window, st.n_cleared, st.n_alive) == \
                   (so.n_observation, so.n_clusters, so.n_marked, so.n_in_window, so.n_cleared, so.n_alive)
            assert _vset(layer.voxels()) == _vset(mo.voxels())
            np.testing.assert_array_equal(layer.lethal(), mo.lethal())
            np.testing.assert_array_equal(layer.dgraph(), mo.dgraph())
            gp, gv = layer.points(with_voxels=True)
            op, ov = mo.points(with_voxels=True)
            assert sorted(map(tuple, np.concatenate([gv, gp], axis=1).tolist())) == sorted(map(tuple, np.concatenate([ov, op], axis=1).tolist()))
        if n_points > 16384 and tol > 0.1:
            assert layer.route_counts()["general"] == 3            # (by choice or by fallback)
        else:
            _check_route(layer, 3)


def test_observation_too_wide_for_the_fused_sort_keys_falls_back(monkeypatch):
    """Points hundreds of metres apart (only a cloud handed over with set_cloud can be): the 0.2 m voxel range does
    not fit the fused route's 28-bit sort keys, the mark phase of that update is redone on the general route --
    results as ever, and the route counters say so."""
    monkeypatch.setenv("DDDMR_MARKING_ROUTE", "auto")
    _, cloud, walls, _ = _scene()
    near = cloud[(np.abs(cloud[:, 0]) < 5.0) & (np.abs(cloud[:, 1]) < 5.0) & (cloud[:, 2] > 0.05) & (cloud[:, 2] < 2.0)][::7][:5000]
    far = np.array([[900.0, 650.0, 40.0, 0], [900.05, 650.0, 40.0, 0], [-700.0, -820.0, -30.0, 0]], dtype=np.float32)
    cfg = marking.shipped_config()
    ground = marking.ground_lattice()
    mo = oracle.MarkingOracle(cfg, ground, walls[:, :3])
    sc = scenes.bench_scene("C2")
    with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
        layer = marking.MarkingLayer(lp, cfg, ground, walls[:, :3])
        for k, obs in enumerate([near, np.concatenate([near[:4000], far]), near[500:4500], np.concatenate([far, near[:3000]])]):
            obs = np.ascontiguousarray(obs, dtype=np.float32)
            t_gb = (0.3 * k, 0.0, 0.0, 0, 0, 0, 1)
            lp.set_cloud(obs)
            st = layer.update(T_BS, t_gb)
            so = mo.update(obs[:, :3], T_BS, t_gb)
            assert (st.n_observation, st.n_clusters, st.n_marked, st.n_in_window, st.n_cleared, st.n_alive) == \
                   (so.n_observation, so.n_clusters, so.n_marked, so.n_in_window, so.n_cleared, so.n_alive)
            assert _vset(layer.voxels()) == _vset(mo.voxels())
            np.testing.assert_array_equal(layer.lethal(), mo.lethal())
            np.testing.assert_array_equal(layer.dgraph(), mo.dgraph())
        rc = layer.route_counts()
        assert rc["fused"] == 2 and rc["general"] == 2, rc


def test_observation_larger_than_the_fused_route_takes_the_general_one(monkeypatch):
    monkeypatch.setenv("DDDMR_MARKING_ROUTE", "auto")
    _, cloud, walls, _ = _scene()
    near = cloud[(cloud[:, 2] > 0.05) & (cloud[:, 2] < 2.0)]
    assert len(near) > 2 * 36000
    cfg = marking.shipped_config(perception_window_size=6.0)
    ground = marking.ground_lattice()
    mo = oracle.MarkingOracle(cfg, ground, walls[:, :3])
    sc = scenes.bench_scene("C2")
    with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
        layer = marking.MarkingLayer(lp, cfg, ground, walls[:, :3])
        inner = near[(np.abs(near[:, 0]) < 5.0) & (np.abs(near[:, 1]) < 5.0)]
        for k, nobs in enumerate([36000, 9000, 33000]):
            obs = np.ascontiguousarray((near if nobs > 32768 else inner)[k::2][:nobs], dtype=np.float32)
            t_gb = (0.2 * k, 0.0, 0.0, 0, 0, 0, 1)
            lp.set_cloud(obs)
            st = layer.update(T_BS, t_gb)
            so = mo.update(obs[:, :3], T_BS, t_gb)
            assert (st.n_observation, st.n_clusters, st.n_marked, st.n_cleared, st.n_alive) == \
                   (so.n_observation, so.n_clusters, so.n_marked, so.n_cleared, so.n_alive)
            assert _vset(layer.voxels()) == _vset(mo.voxels())
            np.testing.assert_array_equal(layer.dgraph(), mo.dgraph())
        rc = layer.route_counts()
        assert rc["fused"] == 1 and rc["general"] == 2, rc


def test_update_next_to_a_pending_tick_gives_the_serial_results():
    """tick_begin -> marking_update -> tick_end (the update on a stream of its own next to the tick's kernels: the
    reference runs the perception thread's doClear_then_Mark and the planner's tick side by side) must leave the store,
    the dGraph, the lethal set and every tick result exactly as set_scan -> marking_update -> tick does; and the update
    is refused when a newer observation was published after tick_begin (it would need a third pinned cloud buffer)."""
    sc, cloud, _, _ = _scene()
    cfg = marking.shipped_config(perception_window_size=10.0)
    ground = marking.ground_lattice()
    name = sc.theory.name.decode()

    def fields(r):
        return (r.planner_state, r.best_index, r.best_cost, r.vx, r.vy, r.wz, r.n_points_binned)
    with LocalPlanner([sc.theory], max_points=1 << 16) as a, LocalPlanner([sc.theory], max_points=1 << 16) as b:
        la = marking.MarkingLayer(a, cfg, ground, np.zeros((0, 3), np.float32))
        lb = marking.MarkingLayer(b, cfg, ground, np.zeros((0, 3), np.float32))
        a.setPlan(sc.plan); b.setPlan(sc.plan)
        for k in range(8):
            t_gb = (0.3 * k, 0.05 * k, 0.0, 0, 0, math.sin(0.02 * k), math.cos(0.02 * k))
            scene_k = cloud if k % 3 else cloud[: len(cloud) // 2]
            scan = scenes.lidar_scan(scene_k, sensor_xyz=(t_gb[0], t_gb[1], 0.5), seed=300 + k)
            a.set_scan(scan, T_BS, t_gb, 10.0, 2.0)
            sa = la.update(T_BS, t_gb)
            ra = a.tick(name, sc.tick)
            b.set_cloud(a.get_cloud())                        # (the feed's centroids come from atomic sums: the same floats for both)
            b.tick_begin(name, sc.tick)
            sb = lb.update(T_BS, t_gb)
            rb = b.tick_end()
            assert fields(ra) == fields(rb)
            assert (sa.n_observation, sa.n_clusters, sa.n_marked, sa.n_in_window, sa.n_cleared, sa.n_alive) == \
                   (sb.n_observation, sb.n_clusters, sb.n_marked, sb.n_in_window, sb.n_cleared, sb.n_alive)
            assert _vset(la.voxels()) == _vset(lb.voxels())
            np.testing.assert_array_equal(la.dgraph(), lb.dgraph())
            np.testing.assert_array_equal(la.lethal(), lb.lethal())
        assert sa.n_alive > 100
        # a newer observation after tick_begin: refused, nothing changes, the tick still ends
        b.tick_begin(name, sc.tick)
        b.set_scan(scenes.lidar_scan(cloud, sensor_xyz=(2.4, 0.4, 0.5), seed=999), T_BS, t_gb, 10.0, 2.0)
        with pytest.raises(RolloutError) as e:
            lb.update(T_BS, t_gb)
        assert e.value.code == K.ERR_STATE
        assert fields(b.tick_end())[:2] == fields(rb)[:2]
        np.testing.assert_array_equal(la.dgraph(), lb.dgraph())


@pytest.mark.parametrize("offset", [(1500.0, -800.0, 30.0), (-4200.5, 3100.25, -12.0)])
def test_marking_far_from_the_map_origin(offset):
    """The C2 scene, the ground nodes and the robot shifted by kilometres (voxel keys of tens of thousands, floats that
    carry 0.1 - 0.5 mm): a ten-scan sequence with a vanishing obstacle, identical to the oracle update by update."""
    sc, cloud, walls, _ = _scene()
    off = np.array(offset, dtype=np.float64)
    sh = lambda a: np.concatenate([(a[:, :3].astype(np.float64) + off).astype(np.float32), a[:, 3:]], axis=1)
    cloud_s, walls_s = sh(cloud), sh(walls)
    ground = (marking.ground_lattice().astype(np.float64) + off).astype(np.float32)
    cfg = marking.shipped_config()
    poses = lambda k: (off[0] + 0.25 * k, off[1] + 0.05 * k, off[2], 0, 0, math.sin(0.03 * k), math.cos(0.03 * k))
    gone = np.hypot(cloud_s[:, 0] - (off[0] + 2.0), cloud_s[:, 1] - off[1]) > 1.2
    scene_of = lambda k, c: cloud_s if k < 5 else cloud_s[gone]
    totals, final = _run_sequence(cfg, walls_s, poses, scene_of, ground=ground, n_updates=10, fragile_tol=1e-5)
    assert totals["marked"] > 500


def test_window_wider_than_the_band_table():
    """A 35 m window spans more rows of ground cells than the node-centric dGraph update has bands (128): the fused route
    then walks every generator point one by one (as the general route always does); five updates with a moving robot,
    identical to the oracle."""
    sc, cloud, walls, _ = _scene()
    cfg = marking.shipped_config(perception_window_size=35.0, inflation_radius=1.2)
    ground = marking.ground_lattice(half=40.0, spacing=0.5)
    poses = lambda k: (0.5 * k, -0.1 * k, 0.0, 0, 0, math.sin(0.05 * k), math.cos(0.05 * k))
    scene_of = lambda k, c: c if k % 2 == 0 else c[np.hypot(c[:, 0] - 3.0, c[:, 1] - 1.0) > 1.5]
    totals, final = _run_sequence(cfg, walls, poses, scene_of, window=35.0, ground=ground, n_updates=5, fragile_tol=1e-5)
    assert totals["marked"] > 1000 and totals["cleared"] > 100


def test_reset_in_the_middle_of_a_sequence():
    """resetdGraph between updates (a new map / a relocalisation): store, dGraph and lethal set start over on both sides and
    the updates that follow are identical again -- the fused route's between-update invariants (zeroed band and cell
    counters, the published counters, the alive list) survive the reset."""
    sc, cloud, walls, _ = _scene()
    cfg = marking.shipped_config()
    ground = marking.ground_lattice()
    mo = oracle.MarkingOracle(cfg, ground, walls[:, :3])
    with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
        layer = marking.MarkingLayer(lp, cfg, ground, walls[:, :3])
        for k in range(9):
            if k in (3, 6):
                layer.reset(); mo.reset()
                assert len(layer.voxels()) == 0
            t_gb = (0.3 * k, 0.05 * k, 0.0, 0, 0, math.sin(0.04 * k), math.cos(0.04 * k))
            scene_k = cloud if k % 2 else cloud[np.hypot(cloud[:, 0] - 2.5, cloud[:, 1]) > 1.0]
            scan = scenes.lidar_scan(scene_k, sensor_xyz=(t_gb[0], t_gb[1], 0.5), seed=700 + k)
            lp.set_scan(scan, T_BS, t_gb, 5.0, 2.0)
            obs = lp.get_cloud()
            st = layer.update(T_BS, t_gb)
            so = mo.update(obs[:, :3], T_BS, t_gb)
            assert (st.n_observation, st.n_clusters, st.n_marked, st.n_in_window, st.n_cleared, st.n_alive) == \
                   (so.n_observation, so.n_clusters, so.n_marked, so.n_in_window, so.n_cleared, so.n_alive), f"update {k}"
            assert _vset(layer.voxels()) == _vset(mo.voxels())
            np.testing.assert_array_equal(layer.lethal(), mo.lethal())
            np.testing.assert_array_equal(layer.dgraph(), mo.dgraph())
