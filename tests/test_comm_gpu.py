"""In-library RCCL exchange (dddmr_rollout_comm_init): torch-free, gloo-free self-test with a
1-rank communicator on the one GPU of the test box -- k_score -> ncclAllReduce(min) of the
(cost bits, -index) slots -> k_resolve on the context's stream must deliver exactly what the
single-rank tick delivers.  (More ranks need more GPUs: RCCL refuses two ranks on one device; the
N > 1 arithmetic is covered by tests/test_sharding_cpu.py and the resolve_words GPU tests.)"""
import numpy as np
import pytest

from dddmr_navigation_amd import _capi as K, configs, scenes
from dddmr_navigation_amd.local_planner import LocalPlanner, RolloutError

pytestmark = pytest.mark.gpu


def _fields(r):
    return (r.planner_state, r.best_index, r.best_cost, r.vx, r.vy, r.wz, r.key, r.n_samples, r.n_local, r.n_points_binned)


@pytest.mark.parametrize("scene", ["playground", "C1", "C2", "blocked", "rotate"])
def test_one_rank_communicator_gives_the_single_rank_result(scene):
    if scene == "playground":
        sc = scenes.playground_scene()
    elif scene == "rotate":
        sc = scenes.bench_scene("C1")
        sc.theory = configs.rotate_inplace_shipped("rot", shortest=True)      # explicit sample list
    else:
        sc = scenes.bench_scene("C1" if scene == "blocked" else scene)
    cloud = sc.cloud if scene != "blocked" else np.array([[0.1, 0.0, 0.3, 0]] * 8, dtype=np.float32)
    name = sc.theory.name.decode()
    with LocalPlanner([sc.theory], max_points=max(len(cloud), 16)) as lp:
        lp.set_cloud(cloud)
        lp.setPlan(sc.plan)
        plain = _fields(lp.tick(name, sc.tick))
        costs = lp.debug()[0].copy()
        lp.comm_init(lp.comm_unique_id(), 0, 1)
        with pytest.raises(RolloutError):
            lp.comm_init(lp.comm_unique_id(), 0, 1)            # already has a communicator
        for _ in range(3):
            assert _fields(lp.tick(name, sc.tick)) == plain
            np.testing.assert_array_equal(lp.debug()[0], costs)
        lp.tick_begin(name, sc.tick)
        assert _fields(lp.tick_end()) == plain
        if plain[1] >= 0:
            assert len(lp.best_poses()) > 0
        lp.comm_destroy()
        assert _fields(lp.tick(name, sc.tick)) == plain
        lp.comm_init(lp.comm_unique_id(), 0, 1)                # and again after a destroy
        assert _fields(lp.tick(name, sc.tick)) == plain
    if scene == "blocked":
        assert plain[0] == K.ALL_TRAJECTORIES_FAIL and plain[1] == -1


def test_comm_init_checks_the_shard():
    sc = scenes.bench_scene("C1")
    with LocalPlanner([sc.theory], rank=1, world_size=2) as lp:
        with pytest.raises(RolloutError) as e:
            lp.comm_init(lp.comm_unique_id(), 0, 2)            # context is rank 1
        assert e.value.code == K.ERR_BAD_ARG
        with pytest.raises(RolloutError):
            lp.comm_init(lp.comm_unique_id(), 1, 3)            # context's world is 2


import os


# DDDMR_COMM_SEEDS=N widens the sweep (default 6)
def test_one_rank_communicator_on_random_scenarios():
    """The random scenarios of tests/test_random_gpu.py through k_score -> ncclAllReduce -> k_resolve (a 1-rank
    communicator, created once: RCCL communicators are not cheap): what the tick delivers must be what the plain tick
    of a second context delivers, field by field."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_random_gpu import random_case
    n = int(os.environ.get("DDDMR_COMM_SEEDS", "6"))
    cases = [random_case(np.random.default_rng(4000 + s + int(os.environ.get("DDDMR_SEED_BASE", "0"))), permute_stack=bool(s & 1))
             for s in range(n)]
    for i, c in enumerate(cases):
        c[0].name = f"t{i % 8}".encode()
    for g in range(0, n, 8):                                   # a context holds a handful of theories
        group = cases[g:g + 8]
        theories = [c[0] for c in group]
        max_pts = max(max(len(c[1]) for c in group), 16)
        with LocalPlanner(theories, max_points=max_pts, max_steps=512) as a, \
                LocalPlanner(theories, max_points=max_pts, max_steps=512) as b:
            b.comm_init(b.comm_unique_id(), 0, 1)
            for th, cloud, plan, tick in group:
                for lp in (a, b):
                    lp.set_cloud(cloud)
                    lp.setPlan(plan)
                name = th.name.decode()
                for _ in range(2):
                    ra, rb = a.tick(name, tick), b.tick(name, tick)
                    assert _fields(ra) == _fields(rb)
                    np.testing.assert_array_equal(a.debug()[0], b.debug()[0])


@pytest.mark.parametrize("world", [2, 3, 4, 7])
@pytest.mark.parametrize("scene", ["rotate", "playground", "C1"])
def test_loopback_exchange_every_rank_resolves_the_global_command(scene, world):
    """The device code a W-rank communicator runs (k_score / k_empty_result -> slot vector -> k_resolve), rehearsed on one
    GPU with dddmr_rollout_comm_loopback: every rank -- including ranks whose shard is EMPTY (the shipped rotate-in-place
    theory has two samples, so with three or more ranks rank 0 owns none) -- must return the unsharded winner and ITS
    command.  (an empty rank decoded the command from sample buffers it never uploaded.)"""
    if scene == "playground":
        sc = scenes.playground_scene()
    else:
        sc = scenes.bench_scene("C1")
        if scene == "rotate":
            sc.theory = configs.rotate_inplace_shipped("rot", shortest=True)
    name = sc.theory.name.decode()
    npts = max(len(sc.cloud), 16)
    with LocalPlanner([sc.theory], max_points=npts) as whole:
        whole.set_cloud(sc.cloud)
        whole.setPlan(sc.plan)
        want = whole.tick(name, sc.tick)
    assert want.best_index >= 0
    ranks = []
    try:
        for r in range(world):
            lp = LocalPlanner([sc.theory], max_points=npts, rank=r, world_size=world)
            lp.set_cloud(sc.cloud)
            lp.setPlan(sc.plan)
            ranks.append(lp)
        words = [lp.winner_words(lp.tick(name, sc.tick)) for lp in ranks]
        assert any(lp.last_result.n_local == 0 for lp in ranks) or scene != "rotate" or world < 3
        for r, lp in enumerate(ranks):
            lp.comm_loopback()
            assert lp.comm_ranks() == world
            for p in range(world):
                if p != r:
                    lp.comm_loopback_set_peer(p, words[p])
            with pytest.raises(RolloutError):
                lp.comm_loopback_set_peer(r, words[r])
            for _ in range(2):
                got = lp.tick(name, sc.tick)
                assert (got.planner_state, got.best_index, got.best_cost) == (want.planner_state, want.best_index, want.best_cost)
                assert (got.vx, got.vy, got.wz) == (want.vx, want.vy, want.wz), f"rank {r} of {world} (n_local {got.n_local})"
            lp.comm_destroy()
            assert lp.comm_ranks() == 0
    finally:
        for lp in ranks:
            lp.close()


@pytest.mark.parametrize("scene", ["C1", "rotate"])
def test_rccl_ranks_on_separate_gpus(scene, tmp_path):
    """The real thing: one process per GPU, dddmr_rollout_comm_init over RCCL, k_score -> ncclAllReduce(min) of the slot
    vector -> k_resolve; every rank must return the unsharded winner and its command.  Needs >= 2 GPUs (RCCL refuses two
    ranks on one device): SKIPPED on the one-GPU test box -- there the same device code runs through the loopback test
    above.  With >= 3 GPUs the rotate scene has an empty shard on rank 0."""
    import subprocess, sys, json
    from dddmr_navigation_amd.local_planner import device_count
    n_gpu = device_count()                                     # (through the library: torch stays out of this process)
    if n_gpu < 2:
        pytest.skip(f"needs >= 2 GPUs for an RCCL communicator with > 1 rank, found {n_gpu}")
    world = min(n_gpu, 3)
    sc = scenes.playground_scene() if scene == "playground" else scenes.bench_scene("C1")
    if scene == "rotate":
        sc.theory = configs.rotate_inplace_shipped("rot", shortest=True)
    with LocalPlanner([sc.theory], max_points=max(len(sc.cloud), 16)) as whole:
        whole.set_cloud(sc.cloud)
        whole.setPlan(sc.plan)
        want = whole.tick(sc.theory.name.decode(), sc.tick)
    helper = os.path.join(os.path.dirname(__file__), "helpers", "comm_rank.py")
    procs = [subprocess.Popen([sys.executable, helper, scene, str(r), str(world), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    try:
        outs = [p.communicate(timeout=240)[0].decode(errors="replace") for p in procs]
    finally:
        for p in procs:                                        # (exactly the children started here)
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{outs[r][-2000:]}"
    n_local = 0
    for r in range(world):
        got = json.load(open(tmp_path / f"rank{r}.json"))
        assert got["comm_ranks"] == world
        for t in got["ticks"]:
            assert (t["state"], t["best_index"], t["best_cost"]) == (want.planner_state, want.best_index, want.best_cost), f"rank {r}"
            assert tuple(t["cmd"]) == (want.vx, want.vy, want.wz), f"rank {r} of {world} (n_local {t['n_local']})"
        n_local += got["ticks"][0]["n_local"]
    assert n_local == want.n_samples


def test_rccl_rank_helper_as_a_single_rank(tmp_path):
    """The child-process helper of the multi-GPU test above, run as a 1-rank world on the box's one GPU, so that the
    helper itself (id hand-over through a file, result file) is exercised wherever the GPU tests run."""
    import subprocess, sys, json
    sc = scenes.bench_scene("C1")
    with LocalPlanner([sc.theory], max_points=max(len(sc.cloud), 16)) as whole:
        whole.set_cloud(sc.cloud)
        whole.setPlan(sc.plan)
        want = whole.tick(sc.theory.name.decode(), sc.tick)
    helper = os.path.join(os.path.dirname(__file__), "helpers", "comm_rank.py")
    p = subprocess.run([sys.executable, helper, "C1", "0", "1", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-2000:]
    got = json.load(open(tmp_path / "rank0.json"))
    assert got["comm_ranks"] == 1
    for t in got["ticks"]:
        assert (t["state"], t["best_index"], t["best_cost"], tuple(t["cmd"])) == (want.planner_state, want.best_index, want.best_cost, (want.vx, want.vy, want.wz))
        assert t["n_local"] == want.n_samples
