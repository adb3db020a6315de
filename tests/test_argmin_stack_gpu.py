"""Order of the critic stack and exactness of the winner, HIP path vs oracle.

The reference runs a theory's critics in `plugins`-array order, whatever that order is
(mpc_critics/src/stacked_scoring_model.cpp:75-93, mpc_critics_ros.cpp:60-81), and picks the
winner by comparing full doubles with `<=` (local_planner.cpp:456-463: equal minima -> the LAST
one; costs above the initial minimum_cost 9999999 are never accepted).  These tests pin
  (a) stacks whose path critics come BEFORE the collision critics (a collided trajectory still
      has to report exactly -1, with every per-trajectory output poisoned before the tick),
  (b) exact ties and near-ties (closer than the packed key's 40 cost bits) on the HIP path,
      unsharded and across shard boundaries,
  (c) the winner against the engine's OWN per-trajectory costs (exact by construction).
"""
import itertools
import os

import numpy as np
import pytest

from dddmr_navigation_amd import _capi as K, configs, scenes, sharding
from dddmr_navigation_amd.local_planner import LocalPlanner
import oracle

pytestmark = pytest.mark.gpu
os.environ["DDDMR_POISON"] = "1"
TOL = 1e-4


def last_argmin(costs):
    """The reference's scan (local_planner.cpp:452-463) over an array of costs."""
    best, m = -1, 9999999.0
    for i, c in enumerate(costs):
        if c >= 0 and c <= m:
            best, m = i, c
    return best


def tick_and_debug(theory, cloud, plan, tick, **kw):
    with LocalPlanner([theory], max_points=max(len(cloud), 16), **kw) as lp:
        lp.set_cloud(cloud)
        lp.setPlan(plan)
        res = lp.tick(theory.name.decode(), tick)
        costs, steps, smp = (a.copy() for a in lp.debug())
    return res, costs, steps, smp


def _critic(kind):
    if kind == K.CRITIC_PURE_PURSUIT:
        return configs.critic(kind, translation_weight=1.0, orientation_weight=0.01)
    return configs.critic(kind, weight=1.0)


PATH = [K.CRITIC_STICK_PATH, K.CRITIC_TOWARD_GLOBAL_PLAN, K.CRITIC_PURE_PURSUIT]
STACKS = []
for coll in ([K.CRITIC_COLLISION], [K.CRITIC_COLLISION_MIN_MAX], [K.CRITIC_COLLISION, K.CRITIC_COLLISION_MIN_MAX]):
    for perm in itertools.permutations(PATH):
        STACKS.append(list(perm) + coll)                       # every path critic ahead of the collision critic(s)
    STACKS.append([PATH[0]] + coll[:1] + [PATH[1]] + coll[1:] + [PATH[2]])   # interleaved
STACKS.append([K.CRITIC_TWIRLING, K.CRITIC_STICK_PATH, K.CRITIC_COLLISION, K.CRITIC_SHORTEST_ANGLE])


@pytest.mark.parametrize("stack", STACKS, ids=lambda s: "-".join(str(k) for k in s))
@pytest.mark.parametrize("scene", ["C1_dd", "C2_omni"])
def test_path_critics_ahead_of_collision(scene, stack):
    if scene == "C1_dd":
        sc = scenes.bench_scene("C1")
        post = np.array([[1.05, 0.45, z, 0.0] for z in np.arange(0.05, 1.0, 0.05)], dtype=np.float32)
        cloud = np.concatenate([sc.cloud, post])
        th = configs.dd_simple_shipped(critics=[_critic(k) for k in stack])
        tick = scenes.tick_input(twist=(0.4, 0.0, 0.1))
    else:
        sc = scenes.bench_scene("C2", "r01")                  # the narrow corridor: most sideways samples hit a wall
        cloud = sc.cloud
        th = configs.omni_simple_shipped(critics=[_critic(k) for k in stack], linear_x_sample=6.0,
                                         linear_y_sample=6.0, angular_z_sample=8.0, sim_time=4.0)
        tick = scenes.tick_input(twist=(0.4, 0.3, 0.0))
    res, costs, steps, smp = tick_and_debug(th, cloud, sc.plan, tick)
    o = oracle.tick(th, cloud, sc.plan, tick, n_threads=8, want_margin=True)
    np.testing.assert_array_equal(steps, o.steps)
    np.testing.assert_array_equal(smp, o.samples)
    assert not np.isnan(costs).any()                                   # nothing left poisoned
    fragile = np.abs(o.min_margin) < TOL
    neg = (costs < 0) | (o.costs < 0)
    assert not (neg & (costs != o.costs) & ~fragile).any()             # reject codes bit-exact: -1 / -4 / -100
    assert (o.costs == -1.0).any() and (o.costs >= 0).any()            # the scene exercises both
    assert set(np.unique(costs[costs < 0])) <= {-1.0, -4.0, -100.0}
    both = (costs >= 0) & (o.costs >= 0)
    assert np.max(np.abs(costs[both] - o.costs[both])) <= TOL
    assert res.best_index == last_argmin(costs)
    if not (neg & (costs != o.costs)).any():
        assert res.planner_state == o.result.planner_state
        if res.best_index != o.result.best_index:                      # libm-level near-tie
            assert abs(costs[res.best_index] - o.costs[o.result.best_index]) <= 1e-6


# ---------------------------------------------------------------------------
# exact ties
# ---------------------------------------------------------------------------
def test_two_identical_samples_higher_index_wins():
    """Rotate-in-place yields (0,0,+w) and (0,0,-w); Twirling scores |w| * weight for both."""
    th = configs.rotate_inplace_shipped("r", critics=[configs.critic(K.CRITIC_TWIRLING)])
    cloud = np.zeros((0, 4), np.float32)
    res, costs, _, _ = tick_and_debug(th, cloud, scenes.s_curve_plan(), scenes.tick_input())
    assert list(costs) == [0.5, 0.5]
    assert (res.best_index, res.best_cost, res.wz) == (1, 0.5, -0.5)
    o = oracle.tick(th, cloud, scenes.s_curve_plan(), scenes.tick_input())
    assert o.result.best_index == 1
    # ... and across a shard boundary: sample 0 on rank 0, sample 1 on rank 1
    slots, keys, lps = [], [], []
    for r in range(2):
        lp = LocalPlanner([th], max_points=16, rank=r, world_size=2)
        lp.set_cloud(cloud)
        lp.setPlan(scenes.s_curve_plan())
        rr = lp.tick("r", scenes.tick_input())
        assert (rr.local_begin, rr.n_local, rr.best_index) == (r, 1, r)
        slots += list(lp.winner_words())
        keys.append(rr.key)
        lps.append(lp)
    for lp in lps:
        w = lp.resolve_words(slots)
        assert (w.best_index, w.best_cost, w.wz, w.planner_state) == (1, 0.5, -0.5, K.TRAJECTORY_FOUND)
        k = lp.resolve(min(keys))
        assert (k.best_index, k.wz) == (1, -0.5)
        lp.close()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_all_equal_costs_last_index_wins(world):
    """Collision critic only: every non-colliding trajectory costs exactly 0 -> the last one wins."""
    sc = scenes.bench_scene("C2")
    th = configs.bench_theory("C2")
    th.n_critics = 1                                     # collision only
    o = oracle.tick(th, sc.cloud, sc.plan, sc.tick, n_threads=8)
    expect = last_argmin(o.costs)
    assert expect == o.result.best_index and 0 < (o.costs == 0).sum() < len(o.costs)
    slots, lps = [], []
    for r in range(world):
        lp = LocalPlanner([th], max_points=len(sc.cloud), rank=r, world_size=world)
        lp.set_cloud(sc.cloud)
        lp.setPlan(sc.plan)
        res = lp.tick(th.name.decode(), sc.tick)
        costs = lp.debug()[0]
        b, e = sharding.shard_range(r, world, 4096)
        np.testing.assert_array_equal(costs, o.costs[b:e])
        la = last_argmin(costs)
        assert res.best_index == (b + la if la >= 0 else -1)
        slots += list(lp.winner_words())
        lps.append(lp)
    for lp in lps:
        w = lp.resolve_words(slots)
        assert (w.best_index, w.best_cost) == (expect, 0.0)
        assert (w.vx, w.vy, w.wz) == (o.result.vx, o.result.vy, o.result.wz)
        lp.close()


# ---------------------------------------------------------------------------
# near-ties below the packed key's resolution (3.7e-9 relative)
# ---------------------------------------------------------------------------
def _near_tie_theory(nx=8, nth=33):
    # cost = {1.0 | 2.0} + |w| * 1e-10: thousands of costs share the key's 40 cost bits but
    # differ as doubles; every operation is exact-or-correctly-rounded on both sides
    return configs.dd_simple_shipped(name="near", linear_x_sample=float(nx), angular_z_sample=float(nth),
                                     acc_lim_theta=100.0,
                                     critics=[configs.critic(K.CRITIC_SHORTEST_ANGLE, weight=1.0),
                                              configs.critic(K.CRITIC_TWIRLING, weight=1e-10)])


@pytest.mark.parametrize("world", [1, 2, 4])
def test_near_ties_resolve_like_full_double_compare(world):
    th = _near_tie_theory()
    cloud = np.zeros((0, 4), np.float32)
    plan = scenes.s_curve_plan()
    tick = scenes.tick_input(twist=(0.4, 0.0, 0.0), heading_deviation=0.5)
    o = oracle.tick(th, cloud, plan, tick)
    n = int(o.result.n_samples)
    assert o.result.best_index == last_argmin(o.costs)
    # the packed key alone would have picked a different (higher) index: the scenario is a real near-tie
    keys = [sharding.pack_key(c, i) for i, c in enumerate(o.costs)]
    assert sharding.key_index(min(keys)) != o.result.best_index
    slots, keys8, lps = [], [], []
    for r in range(world):
        lp = LocalPlanner([th], max_points=16, rank=r, world_size=world)
        lp.set_cloud(cloud)
        lp.setPlan(plan)
        res = lp.tick("near", tick)
        costs = lp.debug()[0]
        b, e = sharding.shard_range(r, world, n)
        np.testing.assert_array_equal(costs, o.costs[b:e])                # no libm in these critics: bit-equal
        la = last_argmin(costs)
        assert res.best_index == b + la and res.best_cost == costs[la]     # exact inside the shard
        slots += list(lp.winner_words())
        keys8.append(res.key)
        lps.append(lp)
    for lp in lps:
        w = lp.resolve_words(slots)
        assert (w.best_index, w.best_cost) == (o.result.best_index, o.result.best_cost)
        assert (w.vx, w.wz) == (o.result.vx, o.result.wz)
        lp.close()


def test_costs_above_the_references_initial_minimum_are_never_accepted():
    """minimum_cost starts at 9999999 (local_planner.cpp:452): a trajectory costing more is not a winner."""
    th = configs.rotate_inplace_shipped("r", critics=[configs.critic(K.CRITIC_TWIRLING, weight=4e7)])
    cloud = np.zeros((0, 4), np.float32)
    res, costs, _, _ = tick_and_debug(th, cloud, scenes.s_curve_plan(), scenes.tick_input())
    assert list(costs) == [2e7, 2e7]
    assert (res.planner_state, res.best_index, res.best_cost, res.wz) == (K.ALL_TRAJECTORIES_FAIL, -1, -1.0, 0.0)
    assert res.key == K.KEY_NONE
    o = oracle.tick(th, cloud, scenes.s_curve_plan(), scenes.tick_input())
    assert (o.result.planner_state, o.result.best_index) == (K.ALL_TRAJECTORIES_FAIL, -1)
    th = configs.rotate_inplace_shipped("r", critics=[configs.critic(K.CRITIC_TWIRLING, weight=2 * 9999999.0)])
    res, costs, _, _ = tick_and_debug(th, cloud, scenes.s_curve_plan(), scenes.tick_input())
    assert list(costs) == [9999999.0, 9999999.0] and res.best_index == 1          # `<=`: the cap itself is accepted


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4"])
def test_winner_is_the_last_exact_minimum_of_the_engines_own_costs(cfg):
    sc = scenes.bench_scene(cfg)
    res, costs, _, _ = tick_and_debug(sc.theory, sc.cloud, sc.plan, sc.tick, max_trajectories=1 << 17)
    la = last_argmin(costs)
    assert res.best_index == la and res.best_cost == costs[la]
    assert res.key == sharding.pack_key(costs[la], la)


# The engine picks its launch shapes and hand-off paths by shard size and by feedback from the previous tick; every
# one of them can also be forced through the environment (read when the context is created).  Whatever is forced, the
# per-trajectory outputs and the winner must be the ones of the default path, bit for bit.
FORCED = [
    {"DDDMR_FINAL": "0"},          # last-workgroup slot reduce, also on a multi-round shard (> 512 slots)
    {"DDDMR_FINAL": "1"},          # k_finalize decode, also on a single-round shard
    {"DDDMR_PROBE": "0"}, {"DDDMR_PROBE": "1"},
    {"DDDMR_RT": "16"}, {"DDDMR_RT": "128"},
    {"DDDMR_THREADS": "256"},
    {"DDDMR_NO_ASSIGN": "1"}, {"DDDMR_NO_TAB": "1"},
]


@pytest.mark.parametrize("cfg", ["C2", "C3"])
def test_forced_launch_shapes_and_hand_off_paths_change_nothing(cfg):
    sc = scenes.bench_scene(cfg)

    def run():
        with LocalPlanner([sc.theory], max_points=len(sc.cloud), max_trajectories=1 << 15) as lp:
            lp.set_cloud(sc.cloud)
            lp.setPlan(sc.plan)
            out = []
            for _ in range(3):           # tick 1: no load feedback, probe on; ticks 2, 3: steady state
                res = lp.tick(sc.theory.name.decode(), sc.tick)
                costs, steps, smp = lp.debug()
                out.append((res.best_index, res.best_cost, (res.vx, res.vy, res.wz), res.key, costs.copy(), steps.copy(), smp.copy()))
            return out

    base = run()
    for env in FORCED:
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            got = run()
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        for (bi, bc, bv, bk, c0, s0, m0), (gi, gc, gv, gk, c1, s1, m1) in zip(base, got):
            assert (gi, gc, gk) == (bi, bc, bk), env
            assert tuple(gv) == tuple(bv), env
            np.testing.assert_array_equal(c1, c0, err_msg=str(env))
            np.testing.assert_array_equal(s1, s0, err_msg=str(env))
            np.testing.assert_array_equal(m1, m0, err_msg=str(env))
