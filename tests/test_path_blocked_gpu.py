"""PathBlockedStrategy::selfMark (dddmr_perception_3d/plugins/path_blocked_strategy.cpp:56-100)
through the C-ABI vs the oracle's kd-tree restatement."""
import numpy as np
import pytest

from dddmr_navigation_amd import _capi as K, configs, host_logic, scenes
from dddmr_navigation_amd.local_planner import LocalPlanner, RolloutError
import oracle

pytestmark = pytest.mark.gpu


def plan_cloud_c2(sc):
    # a global plan = the scene's s-curve, pruned around the robot like prunePlan does
    g = sc.plan
    robot = g[len(g) // 4, :3] + np.array([0.05, -0.03, 0.0])
    pc = host_logic.prune_plan_cloud(g, robot, forward_distance=3.0, backward_distance=1.0)
    assert pc is not None and (pc[:, 3] < 0).any() and (pc[:, 3] > 0).any()
    return pc


@pytest.mark.parametrize("radius", [0.05, 0.3, 0.8, 2.5])
def test_matches_oracle_on_c2(radius):
    sc = scenes.bench_scene("C2")
    pc = plan_cloud_c2(sc)
    with LocalPlanner([sc.theory], max_points=len(sc.cloud)) as lp:
        lp.set_cloud(sc.cloud)
        ratio, op, flags = lp.path_blocked(pc, radius)
    o_ratio, o_op, o_flags = oracle.path_blocked(sc.cloud, pc, radius)
    np.testing.assert_array_equal(flags, o_flags)
    assert ratio == o_ratio and op == o_op
    assert not flags[pc[:, 3] < 0].any()          # backward points are never probed
    if radius >= 0.8:
        assert op == K.OPINION_PATH_BLOCKED_WAIT and flags.any()


# DDDMR_BLOCKED_CASES=N widens the sweep for a soak run (default 12)
def test_random_clouds_and_plans():
    import os
    rng = np.random.default_rng(5)
    th = configs.bench_theory("C1")
    with LocalPlanner([th], max_points=50_000) as lp:
        for case in range(int(os.environ.get("DDDMR_BLOCKED_CASES", "12"))):
            n = int(rng.choice([6, 50, 5000, 40000]))
            cloud = np.zeros((n, 4), np.float32)
            cloud[:, :3] = rng.uniform(-6, 6, (n, 3)) * np.array([1, 1, 0.2])
            m = int(rng.integers(1, 200))
            pc = np.zeros((m, 4), np.float32)
            t = np.linspace(0, 1, m)
            pc[:, 0] = -4 + 8 * t
            pc[:, 1] = 2 * np.sin(3 * t) + rng.uniform(-3, 3)
            pc[:, 2] = rng.uniform(-0.2, 0.2)
            nb = int(rng.integers(0, m))
            pc[:nb, 3] = -1.0
            pc[nb:, 3] = 1.0
            r = float(rng.choice([0.02, 0.1, 0.25, 0.6]))
            if os.environ.get("DDDMR_RANDOM_SHIFT"):         # cloud and plan kilometres from the map origin
                off = np.array([float(v) for v in os.environ["DDDMR_RANDOM_SHIFT"].split(",")])
                cloud[:, :3] = (cloud[:, :3].astype(np.float64) + off).astype(np.float32)
                pc[:, :3] = (pc[:, :3].astype(np.float64) + off).astype(np.float32)
            lp.set_cloud(cloud)
            ratio, op, flags = lp.path_blocked(pc, r)
            o_ratio, o_op, o_flags = oracle.path_blocked(cloud, pc, r)
            np.testing.assert_array_equal(flags, o_flags, err_msg=f"case {case}")
            assert (ratio, op) == (o_ratio, o_op)


def test_edge_cases_and_ratio_arithmetic():
    th = configs.bench_theory("C1")
    pc = np.array([[0, 0, 0, -1], [0.5, 0, 0, 1], [1.0, 0, 0, 1]], np.float32)
    with LocalPlanner([th], max_points=64) as lp:
        # <= 5 observation points: ratio 0, PASS (path_blocked_strategy.cpp:62-64)
        lp.set_cloud(np.array([[0.5, 0, 0, 0]] * 5, np.float32))
        assert lp.path_blocked(pc, 0.3)[:2] == (0.0, K.OPINION_PASS)
        # 6 points on one forward pose: 1 of 3 plan points blocked -> float(1)/float(3)*100.0
        lp.set_cloud(np.array([[0.5, 0, 0, 0]] * 6, np.float32))
        ratio, op, flags = lp.path_blocked(pc, 0.3)
        assert ratio == float(np.float32(1) / np.float32(3)) * 100.0 and op == K.OPINION_PATH_BLOCKED_WAIT
        assert flags.tolist() == [False, True, False]
        # a point exactly at the radius is NOT within it (strict <): 0.25 and 0.0625 are exact in float
        lp.set_cloud(np.array([[0.75, 0, 0, 0]] * 6, np.float32))
        assert lp.path_blocked(pc, 0.25)[1] == K.OPINION_PASS
        assert lp.path_blocked(pc, 0.2500001)[1] == K.OPINION_PATH_BLOCKED_WAIT
        # obstacle only near the backward point: ignored
        lp.set_cloud(np.array([[0.0, 0.05, 0, 0]] * 6, np.float32))
        assert lp.path_blocked(pc, 0.2)[:2] == (0.0, K.OPINION_PASS)
        # empty plan
        assert lp.path_blocked(np.zeros((0, 4), np.float32), 0.3)[:2] == (0.0, K.OPINION_PASS)
        # capacity
        with pytest.raises(RolloutError):
            lp.path_blocked(np.zeros((2000, 4), np.float32), 0.3)
        # the context keeps ticking afterwards
        sc = scenes.bench_scene("C1")
        lp.set_cloud(sc.cloud[:60])
        lp.setPlan(sc.plan)
        assert lp.tick(th.name.decode(), sc.tick).n_samples > 0


def test_state_guards_of_the_query_entry_points():
    """path_blocked / pose arrays refuse to run while a tick_begin is pending, and pose arrays
    need a finished tick; errors leave the context usable."""
    sc = scenes.bench_scene("C1")
    name = sc.theory.name.decode()
    pc = np.array([[0.5, 0, 0, 1]], np.float32)
    with LocalPlanner([sc.theory]) as lp:
        lp.set_cloud(sc.cloud)
        lp.setPlan(sc.plan)
        with pytest.raises(RolloutError) as e:
            lp.pose_arrays()
        assert e.value.code == K.ERR_STATE
        lp.tick_begin(name, sc.tick)
        for call in (lambda: lp.path_blocked(pc, 0.3), lambda: lp.pose_arrays(), lambda: lp.best_poses()):
            with pytest.raises(RolloutError) as e:
                call()
            assert e.value.code == K.ERR_STATE
        res = lp.tick_end()
        assert res.n_samples > 0
        assert len(lp.pose_arrays()) == int(lp.debug()[1].clip(min=0).sum())
        assert lp.path_blocked(pc, 0.3)[1] in (K.OPINION_PASS, K.OPINION_PATH_BLOCKED_WAIT)
