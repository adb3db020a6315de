"""Host logic that stays on the CPU (prunePlan, goal predicate): hand-derived
expectations from local_planner.cpp:374-445, including the duplicated nearest pose."""
import math

import numpy as np

from dddmr_navigation_amd import host_logic
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


def test_prune_plan_cloud_tags_and_order():
    """pcl_prune_plan_ (local_planner.cpp:402-430): backward walk first, in walk order and
    tagged -1; forward walk tagged 1 (0 only for global-plan index 0); the nearest pose
    twice; not reversed, unlike prune_plan_.poses."""
    g = np.zeros((30, 7)); g[:, 0] = np.arange(30) * 0.2; g[:, 6] = 1.0
    pc = host_logic.prune_plan_cloud(g, (2.03, 0.0, 0.0), 1.0, 0.5)
    pr = host_logic.prune_plan(g, (2.03, 0.0, 0.0), 1.0, 0.5)
    assert pc.dtype == np.float32 and pc.shape == (len(pr), 4)
    back = pc[pc[:, 3] < 0]
    fwd = pc[pc[:, 3] >= 0]
    np.testing.assert_allclose(back[:, 0], [2.0, 1.8, 1.6, 1.4], atol=1e-6)       # walk order: idx, idx-1, ...
    np.testing.assert_allclose(fwd[:, 0], np.arange(10, 17) * 0.2, atol=1e-6)
    assert (fwd[:, 3] == 1.0).all()
    np.testing.assert_allclose(pr[:, 0], np.concatenate([back[::-1, 0], fwd[:, 0]]), atol=1e-6)
    # robot at the very start of the plan: the forward walk begins at index 0 -> tag 0
    pc0 = host_logic.prune_plan_cloud(g, (0.01, 0.0, 0.0), 1.0, 0.5)
    assert pc0[0, 3] == -1.0 and pc0[1, 3] == 0.0 and (pc0[2:, 3] == 1.0).all()
    # early returns mirror prune_plan
    assert host_logic.prune_plan_cloud(g[:2], (0, 0, 0), 1.0, 0.5) is None
    assert host_logic.prune_plan_cloud(g, (2.0, 5.0, 0.0), 1.0, 0.5) is None


def _q(yaw, pitch=0.0, roll=0.0):
    from scipy.spatial.transform import Rotation as R
    return tuple(R.from_euler("ZYX", [yaw, pitch, roll]).as_quat())


def test_heading_predicates():
    """getShortestAngleFromPose2RobotHeading / isGoalHeadingAligned / isInitialHeadingAligned
    (local_planner.cpp:198-304) against hand-derived angles and an independent scipy rotation."""
    from scipy.spatial.transform import Rotation as R
    robot = (1.0, 2.0, 0.0) + _q(0.3)
    goal = (5.0, 5.0, 0.0) + _q(1.0)
    assert abs(host_logic.shortest_angle_from_pose_to_robot_heading(robot, goal) - 0.7) < 1e-12
    # wrap-around: robot at +3.0 rad, pose at -3.0 rad -> +0.2832 (not -6.0)
    y = host_logic.shortest_angle_from_pose_to_robot_heading((0, 0, 0) + _q(3.0), (0, 0, 0) + _q(-3.0))
    assert abs(y - (2 * math.pi - 6.0)) < 1e-12
    # tilted robot: yaw of R_b^T R_p by scipy
    rb, rp = R.from_euler("ZYX", [0.4, 0.1, -0.05]), R.from_euler("ZYX", [-1.2, 0.02, 0.03])
    want = (rb.inv() * rp).as_euler("ZYX")[0]
    got = host_logic.shortest_angle_from_pose_to_robot_heading((0, 0, 0) + tuple(rb.as_quat()), (1, 1, 0) + tuple(rp.as_quat()))
    assert abs(got - want) < 1e-12

    g = straight(40, 0.1)
    g[:, 3:7] = _q(0.0)
    aligned, dev = host_logic.is_goal_heading_aligned(g, (0.5, 0, 0) + _q(0.2), yaw_goal_tolerance=0.25)
    assert aligned and abs(dev + 0.2) < 1e-12
    assert host_logic.is_goal_heading_aligned(g, (0.5, 0, 0) + _q(0.3), 0.25)[0] is False
    assert host_logic.is_goal_heading_aligned(np.zeros((0, 7)), robot, 0.25) == (False, None)

    # level plan along +x: the pointing pose has yaw 0 -> deviation = -robot yaw
    aligned, dev = host_logic.is_initial_heading_aligned(g, (0.52, 0.01, 0.0) + _q(0.5), 1.0, 0.3)
    assert not aligned and abs(dev + 0.5) < 1e-12
    assert host_logic.is_initial_heading_aligned(g, (0.52, 0.01, 0.0) + _q(0.1), 1.0, 0.3)[0]
    # plan heading 90 degrees left of the robot
    gy = straight(40, 0.1)[:, [1, 0, 2, 3, 4, 5, 6]]
    aligned, dev = host_logic.is_initial_heading_aligned(gy, (0.0, 0.5, 0.0) + _q(0.0), 1.0, 0.3)
    assert abs(dev - math.pi / 2) < 1e-12
    # sloped plan (vz != 0): axis-angle branch; the pointing pose's x axis is the plan direction
    gs = straight(40, 0.1)
    gs[:, 2] = 0.2 * gs[:, 0]
    aligned, dev = host_logic.is_initial_heading_aligned(gs, (0.5, 0.0, 0.1) + _q(0.25), 1.0, 0.3)
    assert aligned and abs(dev + 0.25) < 1e-9
    # too short a prune plan / off the plan
    assert host_logic.is_initial_heading_aligned(g[:2], robot, 1.0, 0.3) == (False, None)
    assert host_logic.is_initial_heading_aligned(g, (0.5, 3.0, 0.0) + _q(0.0), 1.0, 0.3) == (False, None)
