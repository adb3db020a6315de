"""Theory / critic configurations: the reference's shipped YAML blocks restated
as `TheoryConfig` structs, plus the BASELINE.json bench configurations.

All citations are relative to /root/reference/src/.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Sequence

from . import _capi as K

# Shipped robot cuboid (dddmr_local_planner/local_planner/config/local_planner_play_ground.yaml:13-21)
_CUBOID = {
    "flb": (0.42, 0.36, 0.0), "frb": (0.42, -0.36, 0.0),
    "flt": (0.42, 0.36, 0.6), "frt": (0.42, -0.36, 0.6),
    "blb": (-0.35, 0.36, 0.0), "brb": (-0.35, -0.36, 0.0),
    "blt": (-0.35, 0.36, 0.6), "brt": (-0.35, -0.36, 0.6),
}
# push order of the reference (dd_simple_trajectory_generator_theory.cpp:211-218)
CUBOID_ORDER = ("blb", "brb", "blt", "flb", "brt", "frt", "flt", "frb")


def cuboid_vertices(named=None):
    named = named or _CUBOID
    return [tuple(float(v) for v in named[k]) for k in CUBOID_ORDER]


def critic(kind: int, weight: float = 1.0, translation_weight: float = 0.5,
           orientation_weight: float = 0.5) -> K.CriticConfig:
    """Defaults are the plugins' declare_parameter defaults
    (mpc_critics/models/*.cpp onInitialize)."""
    c = K.CriticConfig()
    c.kind = kind
    c.weight = weight
    c.translation_weight = translation_weight
    c.orientation_weight = orientation_weight
    return c


def theory(name: str, kind: int, critics: Sequence[K.CriticConfig], cuboid=None, **kw) -> K.TheoryConfig:
    """Build a TheoryConfig; unspecified fields take the plugin's
    declare_parameter defaults (dd_simple...cpp:47-134, omni_simple...cpp:47-158,
    dd_rotate_inplace_theory.cpp:47-129)."""
    t = K.TheoryConfig()
    t.name = name.encode()
    t.kind = kind
    defaults = dict(
        use_motor_constraint=0,
        min_vel_x=0.01, max_vel_x=0.1,
        min_vel_y=-0.1, max_vel_y=0.1,
        min_vel_trans=0.0, max_vel_trans=0.1,
        min_vel_theta=0.1, max_vel_theta=0.1,
        acc_lim_x=0.3, acc_lim_y=0.3, acc_lim_theta=0.5,
        deceleration_ratio=2.0,
        max_motor_shaft_rpm=3000.0, wheel_diameter=0.15, gear_ratio=30.0, robot_radius=0.25,
        controller_frequency=10.0, sim_time=2.0,
        linear_x_sample=10.0, linear_y_sample=10.0, angular_z_sample=10.0,
        sim_granularity=0.1, angular_sim_granularity=0.05,
        rotation_speed=0.4,
        bench_fixed_steps=0, bench_no_zero_insert=0,
    )
    defaults.update(kw)
    for k, v in defaults.items():
        if not hasattr(t, k):
            raise KeyError(k)
        setattr(t, k, v)
    verts = cuboid if cuboid is not None else cuboid_vertices()
    assert len(verts) == 8
    for i, v in enumerate(verts):
        for j in range(3):
            t.cuboid[i][j] = float(v[j])
    assert len(critics) <= K.MAX_CRITICS
    t.n_critics = len(critics)
    for i, c in enumerate(critics):
        t.critics[i] = c
    return t


def shipped_dd_critics():
    """dddmr_p2p_move_base/config/p2p_move_base_localization.yaml mpc_critics:
    collision -> stick_path -> pure_pursuit -> toward_global_plan."""
    return [
        critic(K.CRITIC_COLLISION, weight=1.0),
        critic(K.CRITIC_STICK_PATH, weight=0.1),
        critic(K.CRITIC_PURE_PURSUIT, translation_weight=1.0, orientation_weight=0.01),
        critic(K.CRITIC_TOWARD_GLOBAL_PLAN, weight=1.0),
    ]


def dd_simple_shipped(sim_time: float = 2.0, name: str = "differential_drive_simple", **kw) -> K.TheoryConfig:
    """p2p_move_base_localization.yaml:185-214 (sim_time 2.0); the playground
    uses the same block with sim_time 5.0
    (dddmr_local_planner/local_planner/config/local_planner_play_ground.yaml:63-82)."""
    base = dict(
        max_vel_x=1.0, min_vel_x=0.1, max_vel_theta=0.6, min_vel_theta=0.15,
        acc_lim_x=1.0, acc_lim_theta=3.0, deceleration_ratio=2.0,
        max_motor_shaft_rpm=3000.0, wheel_diameter=0.16, gear_ratio=1.0, robot_radius=0.25,
        controller_frequency=10.0, sim_time=sim_time, linear_x_sample=5.0, angular_z_sample=10.0,
        sim_granularity=0.05, angular_sim_granularity=0.025,
    )
    base.update(kw)
    critics = base.pop("critics", None) or shipped_dd_critics()
    return theory(name, K.THEORY_DD_SIMPLE, critics, **base)


def omni_simple_shipped(name: str = "omni_drive_simple", **kw) -> K.TheoryConfig:
    """dddmr_p2p_move_base/config/p2p_wo_mcl.yaml:86-118 + critic stack :122-143
    (shipped DD stack + twirling last)."""
    base = dict(
        max_vel_x=1.0, min_vel_x=-1.0, max_vel_y=1.0, min_vel_y=-1.0,
        max_vel_theta=0.6, min_vel_theta=0.15, min_vel_trans=0.1, max_vel_trans=1.0,
        acc_lim_x=2.0, acc_lim_y=2.0, acc_lim_theta=3.0, deceleration_ratio=2.0,
        use_motor_constraint=0, controller_frequency=10.0, sim_time=2.0,
        linear_x_sample=5.0, linear_y_sample=5.0, angular_z_sample=10.0,
        sim_granularity=0.05, angular_sim_granularity=0.025,
    )
    base.update(kw)
    critics = base.pop("critics", None) or (shipped_dd_critics() + [critic(K.CRITIC_TWIRLING, weight=1.0)])
    return theory(name, K.THEORY_OMNI_SIMPLE, critics, **base)


def rotate_inplace_shipped(name: str = "differential_drive_rotate_inplace", shortest: bool = False, **kw) -> K.TheoryConfig:
    """p2p_move_base_localization.yaml:160-183: DDRotateInplaceTheory with
    rotation_speed 0.5; critics: collision [-> prefer_rotate_shortest]."""
    base = dict(controller_frequency=10.0, rotation_speed=0.5)
    base.update(kw)
    critics = base.pop("critics", None)
    if critics is None:
        critics = [critic(K.CRITIC_COLLISION, weight=1.0)]
        if shortest:
            critics.append(critic(K.CRITIC_SHORTEST_ANGLE, weight=1.0))
    return theory(name, K.THEORY_DD_ROTATE_INPLACE, critics, **base)


def shipped_theories():
    """The three DD theories of the shipped localisation config."""
    return [
        dd_simple_shipped(),
        rotate_inplace_shipped("differential_drive_rotate_inplace"),
        rotate_inplace_shipped("differential_drive_rotate_shortest_angle", shortest=True),
    ]


# ---------------------------------------------------------------------------
# BASELINE.json bench configurations.  Fixed-step mode, exact
# power-of-two sample grids (no inserted zero), shipped DD critic stack.
# ---------------------------------------------------------------------------
BENCH = {
    # name: (theory kind, nx, ny, nth, steps, sim_time, cloud points, seed)
    "C1": dict(kind=K.THEORY_DD_SIMPLE, nx=8, ny=1, nth=8, steps=20, sim_time=2.0, points=5_000, seed=1),
    "C2": dict(kind=K.THEORY_OMNI_SIMPLE, nx=16, ny=16, nth=16, steps=50, sim_time=2.5, points=100_000, seed=2),
    "C3": dict(kind=K.THEORY_OMNI_SIMPLE, nx=32, ny=16, nth=32, steps=80, sim_time=4.0, points=500_000, seed=3),
    "C4": dict(kind=K.THEORY_OMNI_SIMPLE, nx=64, ny=16, nth=64, steps=50, sim_time=2.5, points=100_000, seed=4),
}
BENCH["C3P"] = BENCH["C3"]      # "config 3 also a 10 deg pitch variant" (same theory and cloud, pitched robot pose)


def bench_theory(cfg: str) -> K.TheoryConfig:
    """Theory whose dynamic window is exactly the SURVEY 8d sample box:
    DD: v in [0.1,1.0], w in [-0.6,0.6]; omni: vx,vy in [-1,1], w in [-0.6,0.6].
    Limits are chosen so that initialise() (dd_simple...cpp:236-295,
    omni_simple...cpp:260-332) yields that window for twist (0.5, 0, 0)."""
    b = BENCH[cfg]
    common = dict(
        controller_frequency=10.0, sim_time=b["sim_time"],
        sim_granularity=0.05, angular_sim_granularity=0.025,
        bench_fixed_steps=b["steps"], bench_no_zero_insert=1,
        max_vel_theta=0.6, min_vel_theta=0.0, acc_lim_theta=100.0,
    )
    if b["kind"] == K.THEORY_DD_SIMPLE:
        # min_v = max(0.1, 0.5/5) = 0.1 ; max_v = min(1.0, 0.5 + 100*0.1) = 1.0
        return theory("bench_" + cfg, K.THEORY_DD_SIMPLE, shipped_dd_critics(),
                      min_vel_x=0.1, max_vel_x=1.0, acc_lim_x=100.0, deceleration_ratio=5.0,
                      linear_x_sample=float(b["nx"]), angular_z_sample=float(b["nth"]), **common)
    # omni: deceleration_ratio 1 keeps both deceleration branches off for twist 0.5
    return theory("bench_" + cfg, K.THEORY_OMNI_SIMPLE, shipped_dd_critics(),
                  min_vel_x=-1.0, max_vel_x=1.0, min_vel_y=-1.0, max_vel_y=1.0,
                  min_vel_trans=0.0, max_vel_trans=2.0,
                  acc_lim_x=100.0, acc_lim_y=100.0, deceleration_ratio=1.0,
                  linear_x_sample=float(b["nx"]), linear_y_sample=float(b["ny"]),
                  angular_z_sample=float(b["nth"]), **common)


def theory_array(theories: Iterable[K.TheoryConfig]):
    theories = list(theories)
    arr = (K.TheoryConfig * len(theories))()
    for i, t in enumerate(theories):
        C.memmove(C.byref(arr, i * C.sizeof(K.TheoryConfig)), C.byref(t), C.sizeof(K.TheoryConfig))
    return arr
