"""CPU tests of the C-ABI boundary: the library loads, exports every symbol the
header declares, struct layouts agree, the argmin key orders like the
reference's scan, and the product fails loudly without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dddmr_navigation_amd import _capi as K, configs, sharding
from conftest import ROOT, has_gpu


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dddmr_rollout.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dddmr_rollout_[a-z_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    lib = K.load_library()
    syms = header_symbols()
    assert len(syms) >= 14
    assert sorted(K.EXPORTED_SYMBOLS) == syms
    for s in syms:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.dddmr_rollout_version()


def test_struct_layouts_match_compiled_library():
    lib = K.load_library()
    for i, t in enumerate([K.CriticConfig, K.TheoryConfig, K.RolloutConfig, K.TickInput,
                           K.RolloutResult, K.RolloutDebug, K.MarkingConfig, K.MarkingStats]):
        assert C.sizeof(t) == lib.dddmr_rollout_sizeof(i), t.__name__


def test_planner_state_values_match_reference_enum():
    # dddmr_sys_core/include/dddmr_sys_core/dddmr_enum_states.h:46-54
    from dddmr_navigation_amd import PlannerState
    assert [s.value for s in PlannerState] == list(range(7))
    assert PlannerState.ALL_TRAJECTORIES_FAIL == 2 and PlannerState.TRAJECTORY_FOUND == 4


def test_key_orders_like_reference_scan():
    # min(key) == minimum cost; equal costs -> HIGHEST index (local_planner.cpp:460-463)
    pk = sharding.pack_key
    assert pk(1.0, 5) < pk(2.0, 1)
    assert pk(1.0, 7) < pk(1.0, 3)
    assert pk(0.0, 0) < pk(1e-300, 100)
    assert pk(-1.0, 3) == K.KEY_NONE and pk(float("nan"), 3) == K.KEY_NONE
    assert sharding.key_index(K.KEY_NONE) == -1
    rng = np.random.default_rng(0)
    costs = rng.uniform(0, 10, 2000)
    costs[rng.integers(0, 2000, 300)] = -1.0
    costs[100] = costs[1500] = costs[costs >= 0].min()          # an exact tie
    keys = [pk(c, i) for i, c in enumerate(costs)]
    best = -1
    m = 9999999
    for i, c in enumerate(costs):                                # the reference's loop
        if c >= 0 and c <= m:
            best, m = i, c
    assert sharding.key_index(min(keys)) == best == 1500
    for i in (0, 1, 65535, (1 << 24) - 2):
        assert sharding.key_index(pk(3.25, i)) == i


def test_shard_ranges_partition_contiguously():
    for n in (0, 1, 2, 55, 4096, 65536, 65537):
        for w in (1, 2, 3, 8):
            r = [sharding.shard_range(k, w, n) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))


@pytest.mark.skipif(has_gpu(), reason="this asserts the no-GPU behaviour")
def test_create_fails_loudly_without_gpu():
    from dddmr_navigation_amd.local_planner import LocalPlanner, RolloutError
    with pytest.raises(RolloutError) as e:
        LocalPlanner(configs.shipped_theories())
    assert e.value.code == K.ERR_NO_DEVICE


def test_create_rejects_bad_arguments():
    lib = K.load_library()
    ctx = C.c_void_p()
    assert lib.dddmr_rollout_create(None, C.byref(ctx)) == K.ERR_BAD_ARG
    cfg = K.RolloutConfig()
    cfg.abi_version = 99
    assert lib.dddmr_rollout_create(C.byref(cfg), C.byref(ctx)) == K.ERR_BAD_ARG
    cfg.abi_version = K.ABI_VERSION
    assert lib.dddmr_rollout_create(C.byref(cfg), C.byref(ctx)) == K.ERR_BAD_ARG   # no theories
    assert not ctx.value
    # capacity limits that do not depend on the device
    th = configs.theory_array(configs.shipped_theories())
    cfg.n_theories = len(th)
    cfg.theories = C.cast(th, C.POINTER(K.TheoryConfig))
    cfg.max_points, cfg.max_trajectories, cfg.max_steps, cfg.max_plan_poses = 1 << 20, 1024, 64, 64
    assert lib.dddmr_rollout_create(C.byref(cfg), C.byref(ctx)) == K.ERR_CAPACITY
    cfg.max_points, cfg.max_trajectories = 1024, 1 << 24
    assert lib.dddmr_rollout_create(C.byref(cfg), C.byref(ctx)) == K.ERR_CAPACITY
    cfg.max_trajectories, cfg.max_plan_poses = 1024, 100000
    assert lib.dddmr_rollout_create(C.byref(cfg), C.byref(ctx)) == K.ERR_CAPACITY


def test_shipped_configs_restate_yaml():
    t = configs.dd_simple_shipped()
    assert (t.max_vel_x, t.min_vel_x, t.max_vel_theta, t.sim_time) == (1.0, 0.1, 0.6, 2.0)
    assert [t.critics[i].kind for i in range(t.n_critics)] == [
        K.CRITIC_COLLISION, K.CRITIC_STICK_PATH, K.CRITIC_PURE_PURSUIT, K.CRITIC_TOWARD_GLOBAL_PLAN]
    # cuboid push order blb,brb,blt,flb,... (dd_simple...cpp:211-218)
    assert tuple(round(v, 2) for v in t.cuboid[0]) == (-0.35, 0.36, 0.0)
    assert tuple(round(v, 2) for v in t.cuboid[3]) == (0.42, 0.36, 0.0)
    o = configs.omni_simple_shipped()
    assert o.critics[o.n_critics - 1].kind == K.CRITIC_TWIRLING
    assert configs.bench_theory("C2").bench_fixed_steps == 50
