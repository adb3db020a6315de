"""Randomised differential test: random robot poses (full SE(3)), cuboids (incl.
boxes larger than the 1 m search ball), limits, theories, critic stacks, plans and
clouds -- HIP path vs oracle, same bar as tests/test_parity_gpu.py."""
import math
import os

import numpy as np
import pytest

from dddmr_navigation_amd import _capi as K, configs, scenes, sharding
from dddmr_navigation_amd.local_planner import LocalPlanner
import oracle

pytestmark = pytest.mark.gpu
# every tick starts from poisoned per-trajectory outputs: a trajectory the load-feedback deal
# lost would show up as NaN instead of passing with the previous tick's (identical) values
os.environ["DDDMR_POISON"] = "1"
TOL = 1e-4


# DDDMR_RANDOM_WILD=1 (soak runs): the same generator with the extremes switched on -- vertex lists far from a box
# (jitter up to 0.2 m), degenerate ones (a zero edge, a flat cuboid), tiny cuboids, 100k-point clouds, steep ramps
WILD = os.environ.get("DDDMR_RANDOM_WILD", "0") not in ("", "0")
# DDDMR_RANDOM_SCALE=k multiplies the sample counts per axis (k = 4: up to ~13 000 samples, shards of several rounds
# of k_score workgroups -- the default scenarios all fit one round)
SCALE = int(os.environ.get("DDDMR_RANDOM_SCALE", "1"))


SHIFT = (np.array([float(v) for v in os.environ["DDDMR_RANDOM_SHIFT"].split(",")], dtype=np.float64)
         if os.environ.get("DDDMR_RANDOM_SHIFT") else None)


def random_case(rng, permute_stack=False, wild=None):
    WILD = globals()["WILD"] if wild is None else wild
    kind = rng.choice(["dd", "omni", "rot"], p=[0.45, 0.4, 0.15])
    # cuboid: random box (sometimes long: corners beyond the 1 m ball), reference vertex order
    lx0, lx1 = -rng.uniform(0.1, 0.9), rng.uniform(0.2, 1.3 if rng.random() < 0.3 else 0.7)
    ly = rng.uniform(0.15, 0.6)
    lz0, lz1 = rng.uniform(-0.1, 0.1), rng.uniform(0.3, 1.2)
    named = {"flb": (lx1, ly, lz0), "frb": (lx1, -ly, lz0), "flt": (lx1, ly, lz1), "frt": (lx1, -ly, lz1),
             "blb": (lx0, ly, lz0), "brb": (lx0, -ly, lz0), "blt": (lx0, ly, lz1), "brt": (lx0, -ly, lz1)}
    cub = configs.cuboid_vertices(named)
    if rng.random() < 0.3:
        # not a body-frame box: sheared / jittered vertices take the general vertex path
        # (the collision critic still derives its OBB from vertices 0, 1, 2, 3)
        cub = [tuple(float(c + d) for c, d in zip(v, rng.uniform(-0.05, 0.05, 3))) for v in cub]
    if WILD:
        # (own stream, seeded from the generator's state without drawing from it: the rest of the case stays as it is)
        wr = np.random.default_rng(int(rng.bit_generator.state["state"]["state"]) & 0xFFFFFFFF)
        u = wr.random()
        if u < 0.25:
            cub = [tuple(float(c + d) for c, d in zip(v, wr.uniform(-0.2, 0.2, 3))) for v in cub]
        elif u < 0.30:
            cub = list(cub); cub[1] = cub[0]                               # zero edge v1 - v0
        elif u < 0.35:
            cub = [(v[0], v[1], 0.2) for v in cub]                         # flat: zero edge v2 - v0
        elif u < 0.45:
            cub = [tuple(float(0.12 * c) for c in v) for v in cub]         # a 10 cm robot
    stack = [configs.critic(K.CRITIC_COLLISION if rng.random() < 0.8 else K.CRITIC_COLLISION_MIN_MAX)]
    if rng.random() < 0.2:
        stack.append(configs.critic(K.CRITIC_COLLISION_MIN_MAX))
    pool = [configs.critic(K.CRITIC_STICK_PATH), configs.critic(K.CRITIC_TOWARD_GLOBAL_PLAN, weight=rng.uniform(0.2, 2)),
            configs.critic(K.CRITIC_PURE_PURSUIT, translation_weight=rng.uniform(0.2, 2), orientation_weight=rng.uniform(0, 0.5)),
            configs.critic(K.CRITIC_TWIRLING, weight=rng.uniform(0, 1)), configs.critic(K.CRITIC_SHORTEST_ANGLE, weight=rng.uniform(0.5, 2))]
    for i in rng.permutation(len(pool))[: rng.integers(0, 5)]:
        stack.append(pool[i])
    if permute_stack:
        # any plugin order is legal (mpc_critics_ros.cpp:60-81): shuffle the WHOLE stack, so that path
        # critics also run before the collision critics (a separate generator keeps the rest of the
        # scenario identical to the collision-first variant of the same seed)
        prng = np.random.default_rng(int(rng.integers(1 << 30)) + 77)
        stack = [stack[i] for i in prng.permutation(len(stack))]
    else:
        rng.integers(1 << 30)
    sim_time = float(rng.uniform(1.0, 4.0))
    common = dict(sim_time=sim_time, sim_granularity=float(rng.choice([0.05, 0.1])),
                  angular_sim_granularity=float(rng.choice([0.025, 0.05])), critics=stack, cuboid=cub)
    extra = {}
    if WILD and kind != "rot":
        # the dynamic window's other inputs (dd_simple...cpp:236-295, omni_simple...cpp:260-332): acceleration limits,
        # deceleration ratio, controller frequency, minimum speeds, the motor-shaft constraint (explicit sample list)
        wr2 = np.random.default_rng((int(rng.bit_generator.state["state"]["state"]) >> 7) & 0xFFFFFFFF)
        extra = dict(acc_lim_x=float(wr2.uniform(0.3, 3.0)), acc_lim_theta=float(wr2.uniform(0.5, 4.0)),
                     deceleration_ratio=float(wr2.uniform(1.0, 5.0)), controller_frequency=float(wr2.choice([5.0, 10.0, 20.0])),
                     min_vel_theta=float(wr2.uniform(0.0, 0.3)))
        if kind == "dd":
            extra.update(min_vel_x=float(wr2.uniform(0.0, 0.3)))
            if wr2.random() < 0.4:
                extra.update(use_motor_constraint=1, max_motor_shaft_rpm=float(wr2.uniform(30.0, 200.0)), gear_ratio=1.0,
                             wheel_diameter=float(wr2.uniform(0.1, 0.3)), robot_radius=float(wr2.uniform(0.15, 0.4)))
        else:
            extra.update(acc_lim_y=float(wr2.uniform(0.3, 3.0)), min_vel_trans=float(wr2.uniform(0.0, 0.3)),
                         max_vel_trans=float(wr2.uniform(0.5, 1.5)))
    common.update(extra)
    if kind == "dd":
        th = configs.dd_simple_shipped(name="t", linear_x_sample=float(rng.integers(2, 9) * SCALE),
                                       angular_z_sample=float(rng.integers(2, 14) * SCALE), max_vel_x=float(rng.uniform(0.5, 1.5)),
                                       max_vel_theta=float(rng.uniform(0.3, 1.0)), **common)
        twist = (rng.uniform(0.0, 1.0), 0.0, rng.uniform(-0.3, 0.3))
    elif kind == "omni":
        th = configs.omni_simple_shipped(name="t", linear_x_sample=float(rng.integers(2, 6) * SCALE),
                                         linear_y_sample=float(rng.integers(2, 6) * SCALE),
                                         angular_z_sample=float(rng.integers(2, 9) * SCALE), **common)
        twist = (rng.uniform(-0.5, 0.8), rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3))
    else:
        common.pop("sim_time")
        th = configs.rotate_inplace_shipped("t", rotation_speed=float(rng.uniform(0.2, 0.8)), **common)
        twist = (0.0, 0.0, 0.0)
    # pose: anywhere, any attitude (ramps up to ~20 deg)
    tilt = 0.8 if WILD else 0.35
    q = scenes.quat_from_rpy(rng.uniform(-tilt, tilt), rng.uniform(-tilt, tilt), rng.uniform(-math.pi, math.pi))
    pos = rng.uniform(-30, 30, 3) * np.array([1, 1, 0.1])
    tick = scenes.tick_input(pose=tuple(pos) + q, twist=twist, allowed_max=(-1.0 if rng.random() < 0.7 else rng.uniform(0.2, 1.0)),
                             heading_deviation=rng.uniform(-1, 1))
    # cloud: clutter around the robot (global frame), a few walls, sometimes tiny
    n = int(rng.choice([0, 3, 5, 200, 3000, 20000] + ([100000] if WILD else [])))
    pts = rng.uniform(-5, 5, (n, 3)) * np.array([1, 1, 0.3]) + pos
    if n >= 200:
        wall = np.stack([np.full(400, pos[0] + rng.uniform(0.6, 2.5)), pos[1] + rng.uniform(-3, 3, 400),
                         pos[2] + rng.uniform(-0.5, 1.5, 400)], axis=1)
        pts = np.concatenate([pts, wall])
    cloud = np.zeros((len(pts), 4), dtype=np.float32)
    cloud[:, :3] = pts
    # plan: a curve starting near the robot
    m = int(rng.choice([0, 2, 3, 20, 120]))
    plan = np.zeros((m, 7))
    yaw0 = rng.uniform(-math.pi, math.pi)
    for i in range(m):
        s = 0.05 * i
        plan[i, 0] = pos[0] + s * math.cos(yaw0) + 0.2 * math.sin(s)
        plan[i, 1] = pos[1] + s * math.sin(yaw0)
        plan[i, 2] = pos[2] + 0.02 * s
        plan[i, 3:7] = scenes.quat_from_rpy(0.0, 0.02, yaw0 + 0.3 * math.sin(s))
    if SHIFT is not None:
        # DDDMR_RANDOM_SHIFT="x,y,z": the whole scenario moved by a map-scale offset (a float carries 0.1 - 0.5 mm at
        # kilometres: the reference's float tests then work on big numbers, and so must the device)
        cloud = cloud.copy()
        cloud[:, :3] = (cloud[:, :3].astype(np.float64) + SHIFT).astype(np.float32)
        plan[:, :3] += SHIFT
        for a in range(3):
            tick.robot_pose[a] = float(tick.robot_pose[a]) + float(SHIFT[a])
    return th, cloud, plan, tick


# DDDMR_SEED_BASE shifts every seed of this file (a soak over scenarios the default sweep never sees)
SEED_BASE = int(os.environ.get("DDDMR_SEED_BASE", "0"))


# How often the tolerance branches of this file fire: written to
# gpurun_out/parity_stats_random.json at the end of the module, quoted in DESIGN.md section 5.
STATS = {"runs": 0, "runs_with_fragile_flip": 0, "runs_winner_differs_within_1e-6": 0,
         "runs_winner_identical": 0, "runs_winner_unchecked_after_flip": 0, "max_abs_cost_diff": 0.0}


@pytest.fixture(scope="module", autouse=True)
def _dump_stats():
    yield
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_stats_random.json"), "w") as f:
        json.dump(STATS, f, indent=1)
    print("\n[parity stats, random suite]", json.dumps(STATS))


def _last_argmin(costs):
    best, m = -1, 9999999.0
    for i, c in enumerate(costs):
        if c >= 0 and c <= m:
            best, m = i, c
    return best


# DDDMR_RANDOM_SEEDS=N widens the sweep for a soak run (default 120 keeps the suite short)
@pytest.mark.parametrize("permuted", [False, True], ids=["collision_first", "shuffled_stack"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("DDDMR_RANDOM_SEEDS", "120"))))
def test_random_scenario(seed, permuted):
    rng = np.random.default_rng(1000 + seed + SEED_BASE)
    th, cloud, plan, tick = random_case(rng, permute_stack=permuted)
    with LocalPlanner([th], max_points=max(len(cloud), 16), max_steps=512) as lp:
        lp.set_cloud(cloud)
        lp.setPlan(plan)
        res = lp.tick("t", tick)
        costs, steps, smp = (a.copy() for a in lp.debug())
        # later ticks deal the trajectories to the workgroups by the load the previous tick
        # measured (a different assignment every time): the results must not move by a bit
        for _ in range(3):
            res_n = lp.tick("t", tick)
            costs_n, steps_n, smp_n = lp.debug()
            np.testing.assert_array_equal(costs_n, costs)
            np.testing.assert_array_equal(steps_n, steps)
            np.testing.assert_array_equal(smp_n, smp)
            assert (res_n.best_index, res_n.best_cost, res_n.key) == (res.best_index, res.best_cost, res.key)
    o = oracle.tick(th, cloud, plan, tick, n_threads=8, want_margin=True)
    np.testing.assert_array_equal(steps, o.steps)
    np.testing.assert_array_equal(smp, o.samples)
    fragile = np.abs(o.min_margin) < TOL
    neg = (costs < 0) | (o.costs < 0)
    bad = neg & (costs != o.costs) & ~fragile
    assert not bad.any(), (np.nonzero(bad)[0][:5], costs[bad][:5], o.costs[bad][:5], o.min_margin[bad][:5])
    both = (costs >= 0) & (o.costs >= 0)
    if both.any():
        d = float(np.max(np.abs(costs[both] - o.costs[both])))
        assert d <= TOL
        STATS["max_abs_cost_diff"] = max(STATS["max_abs_cost_diff"], d)
    assert not np.isnan(costs).any()
    # exact by construction: the winner is the last exact minimum of the engine's own costs
    assert res.best_index == _last_argmin(costs)
    if res.best_index >= 0:
        assert res.best_cost == costs[res.best_index]
    STATS["runs"] += 1
    if not (neg & (costs != o.costs)).any():
        # the engine's costs differ from the oracle's by libm-level noise (<= 3e-7): where two of the
        # oracle's costs are closer than that, either sample may legitimately win
        r = o.result
        assert res.planner_state == r.planner_state
        if res.best_index != r.best_index:
            assert abs(costs[res.best_index] - o.costs[r.best_index]) <= 1e-6
            STATS["runs_winner_differs_within_1e-6"] += 1
        else:
            assert abs(res.vx - r.vx) <= TOL and abs(res.vy - r.vy) <= TOL and abs(res.wz - r.wz) <= TOL
            STATS["runs_winner_identical"] += 1
    else:
        STATS["runs_with_fragile_flip"] += 1
        STATS["runs_winner_unchecked_after_flip"] += 1


def test_collision_box_that_sticks_out_of_its_vertices():
    """Found by a soak (scenario 102133): a jittered cuboid -- CollisionModel's box is the mean of the 8 vertices
    +- half the edges v1-v0, v2-v0, v3-v0 (collision_model.cpp:85-115), which for a vertex list that is not a body-frame
    box reaches beyond the vertices' own bounding box.  A cloud point in that sliver collides in the reference; the
    candidate cells used to be taken from the vertices' bounding box and never looked at it."""
    th, cloud, plan, tick = random_case(np.random.default_rng(102133), permute_stack=False, wild=False)
    with LocalPlanner([th], max_points=len(cloud), max_steps=512) as lp:
        lp.set_cloud(cloud)
        lp.setPlan(plan)
        lp.tick("t", tick)
        costs = lp.debug()[0].copy()
    o = oracle.tick(th, cloud, plan, tick, n_threads=8, want_margin=True)
    assert o.costs[38] == -1.0 and abs(o.min_margin[38]) > 5e-3          # a clear collision, 8.7 mm inside the box
    assert costs[38] == -1.0
    np.testing.assert_array_equal(costs < 0, o.costs < 0)


# DDDMR_RANDOM_SHARD_SEEDS=N widens the sweep (default 12)
@pytest.mark.parametrize("seed", range(int(os.environ.get("DDDMR_RANDOM_SHARD_SEEDS", "12"))))
def test_random_scenario_sharded_over_contexts(seed):
    """The random scenarios again, cut into 2..7 contiguous shards (one context each, all on this GPU): the shards'
    per-trajectory outputs tile the unsharded ones bit for bit, and the exact slot-vector resolve
    (dddmr_rollout_winner_words / _resolve_words: minimum cost as full doubles, equal costs -> highest index) yields
    the unsharded winner and command on every rank."""
    rng = np.random.default_rng(1000 + seed + SEED_BASE)
    th, cloud, plan, tick = random_case(rng, permute_stack=bool(seed & 1))
    world = int(np.random.default_rng(7000 + seed).integers(2, 8))
    with LocalPlanner([th], max_points=max(len(cloud), 16), max_steps=512) as lp:
        lp.set_cloud(cloud)
        lp.setPlan(plan)
        res0 = lp.tick("t", tick)
        costs0, steps0, smp0 = (a.copy() for a in lp.debug())
    ctxs, words, parts = [], [], []
    try:
        for r in range(world):
            lp = LocalPlanner([th], max_points=max(len(cloud), 16), max_steps=512, rank=r, world_size=world)
            ctxs.append(lp)
            lp.set_cloud(cloud)
            lp.setPlan(plan)
            res = lp.tick("t", tick)
            b, e = sharding.shard_range(r, world, res.n_samples)
            assert (res.local_begin, res.n_local) == (b, e - b)
            c, s, m = lp.debug()
            parts.append((c[:e - b].copy(), s[:e - b].copy(), m[:e - b].copy()))
            words.append(lp.winner_words(res))
        np.testing.assert_array_equal(np.concatenate([p[0] for p in parts]), costs0[:res0.n_samples])
        np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), steps0[:res0.n_samples])
        np.testing.assert_array_equal(np.concatenate([p[2] for p in parts]), smp0[:res0.n_samples])
        slots = [2 ** 63 - 1] * (2 * world)
        for r, (w0, w1) in enumerate(words):
            slots[2 * r], slots[2 * r + 1] = w0, w1
        for lp in ctxs:
            out = lp.resolve_words(slots)
            assert out.best_index == res0.best_index
            if res0.best_index >= 0:
                assert (out.best_cost, out.vx, out.vy, out.wz) == (res0.best_cost, res0.vx, res0.vy, res0.wz)
            assert out.planner_state == res0.planner_state
    finally:
        for lp in ctxs:
            lp.close()


# DDDMR_RANDOM_SEQ_SEEDS=N widens the sweep (default 6)
@pytest.mark.parametrize("seed", range(int(os.environ.get("DDDMR_RANDOM_SEQ_SEEDS", "6"))))
def test_random_sequences_of_ticks(seed):
    """State carried between ticks must never leak into a result: ONE context with three random theories, ten ticks
    that switch theory at random, move the robot, replace the cloud or the prune plan now and then (load feedback
    from a different theory / pose / cloud, the adaptive probe round, the triple-buffered cloud), each tick against the
    oracle on exactly that tick's inputs."""
    base = 1000 + seed + SEED_BASE
    cases = [random_case(np.random.default_rng(base * 7 + i), permute_stack=bool((seed + i) & 1)) for i in range(3)]
    theories = []
    for i, c in enumerate(cases):
        th = c[0]
        th.name = f"t{i}".encode()
        theories.append(th)
    rng = np.random.default_rng(base + 55)
    max_pts = max(max(len(c[1]) for c in cases), 16)
    cloud, plan = cases[0][1], cases[0][2]
    with LocalPlanner(theories, max_points=max_pts, max_steps=512) as lp:
        lp.set_cloud(cloud)
        lp.setPlan(plan)
        for t in range(10):
            i = int(rng.integers(0, 3))
            th, _, _, tick = cases[i]
            u = rng.random()
            if u < 0.3:
                cloud = cases[int(rng.integers(0, 3))][1]
                if len(cloud) > 10 and rng.random() < 0.5:
                    cloud = cloud[rng.random(len(cloud)) < 0.8]
                lp.set_cloud(cloud)
            elif u < 0.45:
                plan = cases[int(rng.integers(0, 3))][2]
                lp.setPlan(plan)
            # the case's own tick input, the robot nudged along (the cloud and the plan stay where they are)
            pose = np.array([tick.robot_pose[k] for k in range(7)], dtype=np.float64)
            pose[:3] += rng.uniform(-0.3, 0.3, 3) * np.array([1.0, 1.0, 0.05])
            tk = scenes.tick_input(pose=tuple(pose), twist=(tick.robot_twist[0], tick.robot_twist[1], tick.robot_twist[2]),
                                   allowed_max=tick.allowed_max_linear_speed, heading_deviation=tick.heading_deviation)
            res = lp.tick(f"t{i}", tk)
            costs, steps, smp = (a.copy() for a in lp.debug())
            n = res.n_samples
            o = oracle.tick(th, cloud, plan, tk, n_threads=8, want_margin=True)
            np.testing.assert_array_equal(steps[:n], o.steps)
            np.testing.assert_array_equal(smp[:n], o.samples)
            fragile = np.abs(o.min_margin) < TOL
            neg = (costs[:n] < 0) | (o.costs < 0)
            bad = neg & (costs[:n] != o.costs) & ~fragile
            assert not bad.any(), (t, i, np.nonzero(bad)[0][:5], costs[:n][bad][:5], o.costs[bad][:5], o.min_margin[bad][:5])
            both = (costs[:n] >= 0) & (o.costs >= 0)
            if both.any():
                assert float(np.max(np.abs(costs[:n][both] - o.costs[both]))) <= TOL
            assert res.best_index == _last_argmin(costs[:n])
            if not (neg & (costs[:n] != o.costs)).any():
                assert res.planner_state == o.result.planner_state
                if res.best_index != o.result.best_index:
                    assert abs(costs[res.best_index] - o.costs[o.result.best_index]) <= 1e-6


# DDDMR_RANDOM_DEBUG_SEEDS=N widens the sweep (default 8)
@pytest.mark.parametrize("seed", range(int(os.environ.get("DDDMR_RANDOM_DEBUG_SEEDS", "8"))))
def test_random_scenario_debug_outputs(seed):
    """The visualisation outputs of the random scenarios: `trajectory` / `accepted_trajectory` pose arrays
    (local_planner.cpp:549-569, :461-470), the best trajectory's poses and cuboids (:472-478, Trajectory::getCuboid)
    against the oracle's generateTrajectory, sample by sample."""
    rng = np.random.default_rng(1000 + seed + SEED_BASE)
    th, cloud, plan, tick = random_case(rng, permute_stack=bool(seed & 1))
    with LocalPlanner([th], max_points=max(len(cloud), 16), max_steps=512) as lp:
        lp.set_cloud(cloud)
        lp.setPlan(plan)
        res = lp.tick("t", tick)
        costs, steps, smp = (a.copy() for a in lp.debug())
        every = lp.pose_arrays()
        accepted = lp.pose_arrays(accepted_only=True)
        best = lp.best_poses() if res.best_index >= 0 else None
        best_cub = lp.best_cuboids() if res.best_index >= 0 else None
    n = res.n_samples
    want_all, want_acc, per = [], [], {}
    for i in range(n):
        if steps[i] <= 0:
            continue
        ref, ref_cub, _ = oracle.generate(th, tick, smp[i], capacity=1024)
        assert len(ref) == steps[i]
        want_all.append(ref)
        if costs[i] >= 0:                      # (the engine's own verdicts: fragile flips are test_random_scenario's business)
            want_acc.append(ref)
        if i == res.best_index:
            per = dict(ref=ref, cub=ref_cub)
    assert len(every) == int(steps[:n][steps[:n] > 0].sum())
    if want_all:
        np.testing.assert_allclose(every, np.concatenate(want_all), atol=1e-5)
    assert len(accepted) == sum(len(w) for w in want_acc)
    if want_acc:
        np.testing.assert_allclose(accepted, np.concatenate(want_acc), atol=1e-5)
    if res.best_index >= 0:
        np.testing.assert_allclose(best, per["ref"], atol=1e-5)
        assert best_cub.shape == per["cub"].shape
        np.testing.assert_allclose(best_cub, per["cub"], atol=1e-5)
