"""Host-side pieces of the local planner that stay on the CPU (SURVEY.md 8a rows a18
and the heading predicates): tiny, sequential, and their outputs are INPUTS of the
rollout engine.  Restated from
/root/reference/src/dddmr_local_planner/local_planner/src/local_planner.cpp
with the same quirks (cited inline).  Test support for the Python mirror only: in a real
deployment these functions stay what they are in the reference -- unchanged host C++ inside
Local_Planner -- and nothing in the C-ABI or in include/dddmr_rollout.hpp replaces them.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np


def _prune_walk(plan: np.ndarray, robot_xyz, forward_distance: float, backward_distance: float):
    """The two walks of Local_Planner::prunePlan (local_planner.cpp:374-445) as global-plan
    indices: (backward walk in walk order idx, idx-1, ...; forward walk idx, idx+1, ...),
    or None where the reference returns early."""
    if len(plan) < 3:
        return None
    # 1-NN on the float point cloud of the plan with FLANN's float distance (:389)
    pf = plan[:, :3].astype(np.float32)
    q = np.asarray(robot_xyz, dtype=np.float64).astype(np.float32)
    d = pf - q
    d2 = (d[:, 0] * d[:, 0]).astype(np.float32)
    d2 = (d2 + d[:, 1] * d[:, 1]).astype(np.float32)
    d2 = (d2 + d[:, 2] * d[:, 2]).astype(np.float32)
    idx = int(np.argmin(d2))
    if math.sqrt(float(d2[idx])) > 1.0:
        return None

    def dist(a, b):
        return math.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2 + (a[2] - b[2]) ** 2)

    back, fwd = [], []
    last = plan[idx]
    bd = backward_distance
    for i in range(idx, -1, -1):            # backward check (:403-415)
        back.append(i)
        if i < idx:
            bd -= dist(last, plan[i])
        last = plan[i]
        if bd < 0:
            break
    fd = forward_distance
    for i in range(idx, len(plan)):          # forward check (:420-438); last := plan[idx] at i == idx (:435)
        fwd.append(i)
        if i > idx:
            fd -= dist(last, plan[i])
        last = plan[i]
        if fd < 0:
            break
    return back, fwd


def prune_plan(global_plan: np.ndarray, robot_xyz, forward_distance: float, backward_distance: float) -> Optional[np.ndarray]:
    """Local_Planner::prunePlan (local_planner.cpp:374-445).

    global_plan: [G,7] poses (x y z qx qy qz qw).  Returns the prune plan [M,7], or
    None where the reference returns without touching prune_plan_ (fewer than 3
    poses :376-377, or the robot is more than 1 m off the plan :395-399).

    Quirk kept on purpose: the nearest pose is pushed twice (once by the backward
    walk, once by the forward walk, :404 and :421).  Both walks stop at the first
    pose whose accumulated distance EXCEEDS the budget (that pose is included).
    """
    plan = np.ascontiguousarray(global_plan, dtype=np.float64).reshape(-1, 7)
    w = _prune_walk(plan, robot_xyz, forward_distance, backward_distance)
    if w is None:
        return None
    back, fwd = w
    return plan[back[::-1] + fwd].copy()       # std::reverse of the backward part (:417), then the forward part


def prune_plan_cloud(global_plan: np.ndarray, robot_xyz, forward_distance: float, backward_distance: float):
    """pcl_prune_plan_ as Local_Planner::prunePlan fills it next to prune_plan_
    (local_planner.cpp:402-430): [M,4] float32 x y z intensity -- the backward walk's
    points in walk order tagged -1, then the forward walk's points tagged 1 (0 for plan
    index 0).  Unlike prune_plan_.poses it is NOT reversed (:417 reverses the poses only).
    PathBlockedStrategy reads the tag (path_blocked_strategy.cpp:80).  None where the
    reference returns early."""
    plan = np.ascontiguousarray(global_plan, dtype=np.float64).reshape(-1, 7)
    w = _prune_walk(plan, robot_xyz, forward_distance, backward_distance)
    if w is None:
        return None
    back, fwd = w
    out = np.zeros((len(back) + len(fwd), 4), dtype=np.float32)
    out[: len(back), :3] = plan[back, :3]
    out[: len(back), 3] = -1.0
    out[len(back):, :3] = plan[fwd, :3]
    out[len(back):, 3] = [0.0 if i == 0 else 1.0 for i in fwd]
    return out


def is_goal_reached(global_plan: np.ndarray, robot_xyz, xy_goal_tolerance: float) -> bool:
    """Local_Planner::isGoalReached (local_planner.cpp:305-320): 3-D distance to the
    last pose strictly below the tolerance."""
    plan = np.asarray(global_plan, dtype=np.float64).reshape(-1, 7)
    if len(plan) == 0:
        return False
    d = np.asarray(robot_xyz, dtype=np.float64) - plan[-1, :3]
    return bool(xy_goal_tolerance > math.sqrt(float(d @ d)))


# ---- heading predicates (they produce ModelSharedData::heading_deviation_, an input of the tick) ----

def _quat_to_matrix(q):
    """tf2::Matrix3x3::setRotation(q), q = (x, y, z, w)."""
    x, y, z, w = (float(v) for v in q)
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return np.array([[1.0 - (yy + zz), xy - wz, xz + wy],
                     [xy + wz, 1.0 - (xx + zz), yz - wx],
                     [xz - wy, yz + wx, 1.0 - (xx + yy)]])


def _normalize_angle(a: float) -> float:
    """angles::normalize_angle"""
    r = math.fmod(a + math.pi, 2.0 * math.pi)
    return r + math.pi if r <= 0.0 else r - math.pi


def shortest_angle_from_pose_to_robot_heading(robot_pose, pose) -> float:
    """Local_Planner::getShortestAngleFromPose2RobotHeading (local_planner.cpp:198-216):
    yaw (tf2 getRPY, solution 1) of inverse(T_gbl_base) * pose, through
    angles::shortest_angular_distance(0, yaw).  Poses are x y z qx qy qz qw."""
    rb = _quat_to_matrix(robot_pose[3:7])
    rp = _quat_to_matrix(pose[3:7])
    m = rb.T @ rp                                   # rotation of inverse(base) * pose
    if abs(m[2, 0]) >= 1.0:
        yaw = 0.0                                   # getEulerYPR's gimbal-lock branch
    else:
        pitch = -math.asin(m[2, 0])
        yaw = math.atan2(m[1, 0] / math.cos(pitch), m[0, 0] / math.cos(pitch))
    return _normalize_angle(yaw - 0.0)


def is_goal_heading_aligned(global_plan: np.ndarray, robot_pose, yaw_goal_tolerance: float):
    """Local_Planner::isGoalHeadingAligned (local_planner.cpp:271-304) ->
    (aligned, heading_deviation); (False, None) for an empty plan (heading_deviation_ untouched)."""
    plan = np.asarray(global_plan, dtype=np.float64).reshape(-1, 7)
    if len(plan) == 0:
        return False, None
    yaw = shortest_angle_from_pose_to_robot_heading(robot_pose, plan[-1])
    return abs(yaw) < yaw_goal_tolerance, yaw


def is_initial_heading_aligned(global_plan: np.ndarray, robot_pose, heading_tracking_distance: float,
                               heading_align_angle: float):
    """Local_Planner::isInitialHeadingAligned (local_planner.cpp:218-269) ->
    (aligned, heading_deviation).  prunePlan(heading_tracking_distance, 0.0); fewer than 3
    prune poses -> (False, None).  The pointing pose sits at the first prune pose and looks at
    the last one: a planar yaw when the two are level (vz == 0, :247-252), otherwise the
    rotation about axis x up by -acos(axis . up) with up = +x (:234-245; `right_vector.normalized()`
    discards its result there, the quaternion constructor normalises the axis anyway)."""
    plan = np.ascontiguousarray(global_plan, dtype=np.float64).reshape(-1, 7)
    pr = prune_plan(plan, robot_pose[:3], heading_tracking_distance, 0.0)
    if pr is None:
        # prunePlan returned early and left prune_plan_ as it was; with no earlier plan that is empty
        return False, None
    if len(pr) < 3:
        return False, None
    first, last = pr[0], pr[-1]
    vx, vy, vz = (float(v) for v in (last[:3] - first[:3]))
    if vz != 0:
        unit = math.sqrt(vx * vx + vy * vy + vz * vz)
        axis = np.array([vx / unit, vy / unit, vz / unit])
        up = np.array([1.0, 0.0, 0.0])
        right = np.cross(axis, up)
        angle = -1.0 * math.acos(float(axis @ up))
        d = math.sqrt(float(right @ right))          # tf2::Quaternion(axis, angle): setRotation
        s = math.sin(angle * 0.5) / d
        q = np.array([right[0] * s, right[1] * s, right[2] * s, math.cos(angle * 0.5)])
        q = q / math.sqrt(float(q @ q))              # q_pre.normalize()
    else:
        yaw = math.atan2(vy, vx)                     # setRPY(0, 0, yaw)
        q = np.array([0.0, 0.0, math.sin(yaw * 0.5), math.cos(yaw * 0.5)])
    pose = np.concatenate([first[:3], q])
    yaw = shortest_angle_from_pose_to_robot_heading(robot_pose, pose)
    return abs(yaw) < heading_align_angle, yaw
