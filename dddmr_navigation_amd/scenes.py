"""Deterministic synthetic scenes for the BASELINE.json configurations
(SURVEY.md 8d): ground-removed obstacle clouds like LeGO-LOAM's
`segmented_cloud_pure`, surfaces sampled on a jittered 5 cm lattice, plus the
S-curve prune plan.  numpy Generator(PCG64(seed)) -- same image here and on the
GPU box, so the bytes are identical on both.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import _capi as K
from . import configs


@dataclass
class Scene:
    name: str
    theory: K.TheoryConfig
    cloud: np.ndarray   # [P,4] float32 x y z intensity (global frame)
    plan: np.ndarray    # [M,7] float64 x y z qx qy qz qw
    tick: K.TickInput


def tick_input(pose=(0, 0, 0, 0, 0, 0, 1), twist=(0.5, 0.0, 0.0), allowed_max=-1.0,
               heading_deviation=0.0) -> K.TickInput:
    t = K.TickInput()
    t.robot_pose[:] = [float(v) for v in pose]
    t.robot_twist[:] = [float(v) for v in twist]
    t.allowed_max_linear_speed = float(allowed_max)
    t.heading_deviation = float(heading_deviation)
    return t


def quat_from_rpy(roll: float, pitch: float, yaw: float):
    cr, sr = math.cos(roll / 2), math.sin(roll / 2)
    cp, sp = math.cos(pitch / 2), math.sin(pitch / 2)
    cy, sy = math.cos(yaw / 2), math.sin(yaw / 2)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
            cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)


def s_curve_plan(m: int = 80, spacing: float = 0.05, x0: float = -1.0) -> np.ndarray:
    """M poses, 5 cm spacing in x, y = 0.5 sin(0.5 x), orientation tangent;
    1 m behind ... 3 m ahead of the origin."""
    plan = np.zeros((m, 7), dtype=np.float64)
    for i in range(m):
        x = x0 + spacing * i
        yaw = math.atan(0.25 * math.cos(0.5 * x))
        plan[i, 0] = x
        plan[i, 1] = 0.5 * math.sin(0.5 * x)
        plan[i, 3:7] = quat_from_rpy(0.0, 0.0, yaw)
    return plan


def straight_plan(goal_xy, m: int = 20) -> np.ndarray:
    """The playground's plan: m poses from the origin towards goal, default
    orientation (local_planner_play_ground_node.cpp:225-233; a default
    geometry_msgs Quaternion is x=y=z=0, w=1)."""
    plan = np.zeros((m, 7), dtype=np.float64)
    dx, dy = goal_xy[0] / m, goal_xy[1] / m
    for i in range(m):
        plan[i, 0] = dx * i
        plan[i, 1] = dy * i
        plan[i, 6] = 1.0
    return plan


# ---------------------------------------------------------------------------
# surface samplers (5 cm lattice, +-1 cm jitter)
# ---------------------------------------------------------------------------
_LAT = 0.05
_JIT = 0.01


def _jitter(rng, pts):
    return pts + rng.uniform(-_JIT, _JIT, size=pts.shape)


def _wall_y(rng, y, x0, x1, z0, z1, lat=_LAT):
    xs = np.arange(x0, x1, lat)
    zs = np.arange(z0 + lat / 2, z1, lat)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    pts = np.stack([X.ravel(), np.full(X.size, y), Z.ravel()], axis=1)
    return _jitter(rng, pts)


def _pillar(rng, cx, cy, r, z0, z1, lat=_LAT):
    n = max(8, int(round(2 * math.pi * r / lat)))
    ang = np.arange(n) * (2 * math.pi / n)
    zs = np.arange(z0 + lat / 2, z1, lat)
    A, Z = np.meshgrid(ang, zs, indexing="ij")
    pts = np.stack([cx + r * np.cos(A.ravel()), cy + r * np.sin(A.ravel()), Z.ravel()], axis=1)
    return _jitter(rng, pts)


def _box(rng, cx, cy, sx, sy, yaw, z0, z1, lat=_LAT):
    """vertical faces of an sx x sy box rotated by yaw about z"""
    faces = []
    zs = np.arange(z0 + lat / 2, z1, lat)
    for (ax, half, other) in ((0, sx / 2, sy / 2), (1, sy / 2, sx / 2)):
        us = np.arange(-other, other + 1e-9, lat)
        U, Z = np.meshgrid(us, zs, indexing="ij")
        for sgn in (-1.0, 1.0):
            if ax == 0:
                lx, ly = np.full(U.size, sgn * half), U.ravel()
            else:
                lx, ly = U.ravel(), np.full(U.size, sgn * half)
            faces.append(np.stack([lx, ly, Z.ravel()], axis=1))
    loc = np.concatenate(faces, axis=0)
    c, s = math.cos(yaw), math.sin(yaw)
    pts = np.stack([cx + c * loc[:, 0] - s * loc[:, 1], cy + s * loc[:, 0] + c * loc[:, 1], loc[:, 2]], axis=1)
    return _jitter(rng, pts)


def _slab(rng, z, x0, x1, y0, y1, lat):
    xs = np.arange(x0, x1, lat)
    ys = np.arange(y0, y1, lat)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), np.full(X.size, z)], axis=1)
    return _jitter(rng, pts)


def _plan_distance(plan_xy, x, y):
    d = np.hypot(plan_xy[:, 0] - x, plan_xy[:, 1] - y)
    return float(d.min())


def _finish(rng, parts, n_points):
    pts = np.concatenate(parts, axis=0)
    if len(pts) < n_points:
        raise ValueError(f"scene generator produced {len(pts)} < {n_points} points")
    sel = rng.permutation(len(pts))[:n_points]
    sel.sort()
    out = np.zeros((n_points, 4), dtype=np.float32)
    out[:, :3] = pts[sel].astype(np.float32)
    return out


def _corridor_floor(rng, plan, z_off, n_obstacles, n_blocking, half_window=10.0, wall_y=1.8,
                    height=2.0, clear_radius=1.1):
    """One floor: corridor walls y = +-wall_y over the 20 m window plus random
    pillars / boxes; centres >= 0.6 m from the plan except n_blocking ones, and
    never within clear_radius of the origin (the robot must start collision-free;
    a larger radius thins the obstacles the sampled trajectories can reach)."""
    parts = [
        _wall_y(rng, wall_y, -half_window, half_window, z_off, z_off + height),
        _wall_y(rng, -wall_y, -half_window, half_window, z_off, z_off + height),
    ]
    plan_xy = plan[:, :2]
    placed = 0
    blocking = 0
    guard = 0
    while placed < n_obstacles and guard < 10000:
        guard += 1
        want_block = blocking < n_blocking
        if want_block:
            cx = rng.uniform(1.6, 3.4)
            cy = 0.5 * math.sin(0.5 * cx) + rng.uniform(-0.45, 0.45)
        else:
            cx = rng.uniform(-half_window + 0.5, half_window - 0.5)
            cy = rng.uniform(-wall_y + 0.3, wall_y - 0.3)
        if math.hypot(cx, cy) < (clear_radius if not want_block else 1.1):
            continue
        if not want_block and _plan_distance(plan_xy, cx, cy) < 0.6 + 0.35:
            continue
        if rng.uniform() < 0.5:
            parts.append(_pillar(rng, cx, cy, rng.uniform(0.10, 0.30), z_off, z_off + height))
        else:
            parts.append(_box(rng, cx, cy, rng.uniform(0.3, 0.9), rng.uniform(0.3, 0.9),
                              rng.uniform(0, math.pi), z_off, z_off + height))
        placed += 1
        blocking += 1 if want_block else 0
    return parts


def cloud_c1(seed: int = 1, n_points: int = 5_000) -> np.ndarray:
    """C1: two walls y = +-1.5 m, x in [-2,8], z in [0,1.5] + 3 pillars r = 0.15."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = [
        _wall_y(rng, 1.5, -2.0, 8.0, 0.0, 1.5),
        _wall_y(rng, -1.5, -2.0, 8.0, 0.0, 1.5),
        _pillar(rng, 2.0, 0.9, 0.15, 0.0, 1.5),
        _pillar(rng, 3.2, -0.7, 0.15, 0.0, 1.5),
        _pillar(rng, 1.4, -1.0, 0.15, 0.0, 1.5),
    ]
    return _finish(rng, parts, n_points)


# Scene layouts.  "r02" (default) is tuned so that about a quarter of the sampled trajectories
# collide, as SURVEY.md 8d asks ("target ~25 % colliding trajectories"): with the corridor walls at
# +-1.8 m, every sample with |vy| * sim_time beyond ~1.4 m ends in a wall, which made 69 % (C2) and
# 86 % (C3) of the round-1 scenes' trajectories collide -- and colliding trajectories are the cheap
# ones (early exit, no path critics).  "r01" keeps the round-1 layouts for comparison.
LAYOUTS = {
    "r02": {"C2": dict(wall_y=4.5, n_obstacles=40, n_blocking=2, clear_radius=1.1),
            "C3": dict(wall_y=6.0, n_obstacles=48, n_blocking=2, clear_radius=3.7, n_obstacles_other=60)},
    "r01": {"C2": dict(wall_y=1.8, n_obstacles=40, n_blocking=6, clear_radius=1.1),
            "C3": dict(wall_y=1.8, n_obstacles=48, n_blocking=6, clear_radius=1.1, n_obstacles_other=48)},
}
DEFAULT_LAYOUT = "r02"


def cloud_c2(seed: int = 2, n_points: int = 100_000, layout: str = DEFAULT_LAYOUT) -> np.ndarray:
    """C2: 20 x 20 m window, corridor walls, 40 pillars/boxes (some blocking the plan),
    z in [0,2]; the window's outer walls fill up the point budget."""
    L = LAYOUTS[layout]["C2"]
    rng = np.random.Generator(np.random.PCG64(seed))
    plan = s_curve_plan()
    parts = _corridor_floor(rng, plan, 0.0, L["n_obstacles"], L["n_blocking"], wall_y=L["wall_y"],
                            clear_radius=L["clear_radius"])
    # outer walls of the window (beyond the corridor, never reachable)
    parts += [_wall_y(rng, 9.9, -10, 10, 0.0, 2.0), _wall_y(rng, -9.9, -10, 10, 0.0, 2.0)]
    return _finish(rng, parts, n_points)


def cloud_c3(seed: int = 3, n_points: int = 500_000, layout: str = DEFAULT_LAYOUT) -> np.ndarray:
    """C3: same footprint, three floors (z offsets 0, 3, 6 m; different layouts)
    and the ceiling slabs' undersides at z = 2.6 and 5.6 m."""
    L = LAYOUTS[layout]["C3"]
    rng = np.random.Generator(np.random.PCG64(seed))
    plan = s_curve_plan()
    parts = []
    for k, z_off in enumerate((0.0, 3.0, 6.0)):
        parts += _corridor_floor(rng, plan, z_off, L["n_obstacles"] if k == 0 else L["n_obstacles_other"],
                                 L["n_blocking"] if k == 0 else 0, wall_y=L["wall_y"],
                                 clear_radius=L["clear_radius"] if k == 0 else 1.1)
        parts += [_wall_y(rng, 9.9, -10, 10, z_off, z_off + 2.0), _wall_y(rng, -9.9, -10, 10, z_off, z_off + 2.0)]
    parts += [_slab(rng, 2.6, -10, 10, -10, 10, 0.08), _slab(rng, 5.6, -10, 10, -10, 10, 0.08)]
    return _finish(rng, parts, n_points)


def bench_scene(cfg: str, layout: str = DEFAULT_LAYOUT) -> Scene:
    """Scene for BASELINE.json config C1..C4 (C4 = C2's cloud, 65536 samples; C3P = C3 with the robot pitched 10 degrees)."""
    b = configs.BENCH[cfg]
    if cfg == "C1":
        cloud = cloud_c1(b["seed"], b["points"])
    elif cfg in ("C3", "C3P"):
        cloud = cloud_c3(b["seed"], b["points"], layout)
        if cfg == "C3P":        # the robot on a 10 degree ramp: cuboids tilted out of the grid's axes, poses climbing in z
            return Scene(cfg, configs.bench_theory(cfg), cloud, s_curve_plan(),
                         tick_input(pose=(0.0, 0.0, 0.0) + tuple(quat_from_rpy(0.0, math.radians(10.0), 0.0))))
    else:
        cloud = cloud_c2(configs.BENCH["C2"]["seed"] if layout != "r01" else b["seed"], b["points"], layout)
    return Scene(cfg, configs.bench_theory(cfg), cloud, s_curve_plan(), tick_input())


def playground_scene(goal=(3.0, 1.0), sim_time: float = 5.0) -> Scene:
    """Fixture F1 = the reference's only fixed-input scenario
    (local_planner_play_ground_node.cpp:206-298, config
    local_planner_play_ground.yaml:63-125): robot at identity, v = 0.4, 5
    obstacle points near (0.8, 0.6, 0.2), 20-pose straight plan to `goal`."""
    cloud = np.array([[0.80, 0.60, 0.2, 0], [0.75, 0.65, 0.2, 0], [0.85, 0.55, 0.2, 0],
                      [0.70, 0.70, 0.2, 0], [0.90, 0.50, 0.2, 0]], dtype=np.float32)
    return Scene("playground", configs.dd_simple_shipped(sim_time=sim_time), cloud,
                 straight_plan(goal), tick_input(twist=(0.4, 0.0, 0.0)))


def lidar_scan(cloud_xyz: np.ndarray, sensor_xyz=(0.0, 0.0, 0.5), seed: int = 0, rings: int = 16,
               azimuths: int = 1800, fov_deg: float = 15.0, sigma: float = 0.01,
               max_range: float = 30.0) -> np.ndarray:
    """C5: the scene seen by a 16-ring, 1800-azimuth spinning LiDAR at
    sensor_xyz (sensor axes = global axes), +-15 deg vertical FOV, gaussian
    range noise.  Spherical z-buffer over the scene cloud: per (ring, azimuth)
    bin the nearest point survives.  Returns [K,3] float32 in the SENSOR frame."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rel = cloud_xyz[:, :3].astype(np.float64) - np.asarray(sensor_xyz, dtype=np.float64)
    rng_xy = np.hypot(rel[:, 0], rel[:, 1])
    dist = np.sqrt(rng_xy ** 2 + rel[:, 2] ** 2)
    el = np.degrees(np.arctan2(rel[:, 2], rng_xy))
    az = np.arctan2(rel[:, 1], rel[:, 0])
    ok = (np.abs(el) <= fov_deg) & (dist > 0.3) & (dist < max_range)
    rel, dist, el, az = rel[ok], dist[ok], el[ok], az[ok]
    ring = np.clip(np.round((el + fov_deg) / (2 * fov_deg) * (rings - 1)).astype(np.int64), 0, rings - 1)
    # a beam only returns if the point is close to the ring's elevation
    ring_el = -fov_deg + ring * (2 * fov_deg / (rings - 1))
    near = np.abs(el - ring_el) <= 0.5
    azb = np.floor((az + np.pi) / (2 * np.pi) * azimuths).astype(np.int64) % azimuths
    key = ring * azimuths + azb
    key, rel, dist = key[near], rel[near], dist[near]
    order = np.lexsort((dist, key))
    key, rel, dist = key[order], rel[order], dist[order]
    first = np.ones(len(key), dtype=bool)
    first[1:] = key[1:] != key[:-1]
    rel, dist = rel[first], dist[first]
    noisy = dist + rng.normal(0.0, sigma, size=dist.shape)
    pts = rel * (noisy / dist)[:, None]
    return pts.astype(np.float32)
