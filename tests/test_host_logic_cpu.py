"""Host logic that stays on the CPU (prunePlan, goal predicate): hand-derived
expectations from local_planner.cpp:374-445, including the duplicated nearest pose."""
import numpy as np

from dddmr_navigation_amd.host_logic import prune_plan, is_goal_reached


def straight(n, step=0.1):
    p = np.zeros((n, 7))
    p[:, 0] = np.arange(n) * step
    p[:, 6] = 1.0
    return p


def test_prune_plan_walks_and_duplicates_nearest_pose():
    plan = straight(100)                                 # x = 0 .. 9.9
    out = prune_plan(plan, (5.02, 0.0, 0.0), forward_distance=3.0, backward_distance=1.0)
    xs = np.round(out[:, 0], 3)
    # nearest pose x=5.0 appears twice (pushed by both walks)
    assert (xs == 5.0).sum() == 2
    # backward: poses until the accumulated distance exceeds 1.0 -> 5.0 .. 3.9 (12 poses)
    assert xs[0] == 3.9
    # forward: until the accumulated distance exceeds 3.0 -> 5.0 .. 8.1
    assert xs[-1] == 8.1
    assert np.all(np.diff(xs) >= 0)


def test_prune_plan_no_update_cases():
    assert prune_plan(straight(2), (0, 0, 0), 3.0, 1.0) is None          # fewer than 3 poses
    assert prune_plan(straight(50), (2.0, 1.5, 0.0), 3.0, 1.0) is None   # > 1 m off the plan
    out = prune_plan(straight(50), (2.0, 0.99, 0.0), 3.0, 1.0)
    assert out is not None


def test_prune_plan_at_plan_ends():
    plan = straight(30)
    out = prune_plan(plan, (0.0, 0.0, 0.0), 3.0, 1.0)       # at the start: nothing behind
    assert out[0, 0] == 0.0 and (np.round(out[:, 0], 3) == 0.0).sum() == 2
    out = prune_plan(plan, (2.9, 0.0, 0.0), 3.0, 1.0)       # at the end: nothing ahead
    assert round(out[-1, 0], 3) == 2.9


def test_goal_reached_is_strict_3d():
    plan = straight(10)
    assert is_goal_reached(plan, (0.9, 0.0, 0.29), 0.3)
    assert not is_goal_reached(plan, (0.9, 0.0, 0.3), 0.3)
    assert not is_goal_reached(plan[:0], (0, 0, 0), 0.3)
