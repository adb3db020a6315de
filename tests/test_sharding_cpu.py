"""world_size-2 gloo test of the N>1 path: contiguous shards scored
independently, ONE 8-byte min all-reduce of the packed key, winner resolved on
every rank == the unsharded result (including the last-wins tie-break across the
shard boundary)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dddmr_navigation_amd import scenes, configs, sharding, _capi as K
import oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(which):
    if which == "C1":
        return scenes.bench_scene("C1")
    if which == "tie":
        # two identical samples (+w, -w twirl cost) living on different ranks
        sc = scenes.playground_scene()
        sc.theory = configs.rotate_inplace_shipped("r", critics=[configs.critic(K.CRITIC_TWIRLING)])
        return sc
    return scenes.playground_scene()


def _worker(rank, world, port, which, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = _scene(which)
        n = len(oracle.samples(sc.theory, sc.tick))
        b, e = sharding.shard_range(rank, world, n)
        o = oracle.tick(sc.theory, sc.cloud, sc.plan, sc.tick, begin=b, end=e)   # the checker scores the shard
        key = sharding.pack_key(o.result.best_cost, max(o.result.best_index, 0))
        red = sharding.all_reduce_key(key)
        # exact form: (cost bits, -index) slots, one min all-reduce of 2*world words
        slots = sharding.all_reduce_words(sharding.winner_words(o.result.best_cost, o.result.best_index), rank, world)
        q.put((rank, b, e, key, red, sharding.reduce_words(slots)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["playground", "C1", "tie"])
def test_two_rank_argmin_equals_unsharded(which):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, which, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sc = _scene(which)
    full = oracle.tick(sc.theory, sc.cloud, sc.plan, sc.tick)
    reduced = {o[4] for o in out}
    assert len(reduced) == 1                       # every rank holds the same winner
    assert sharding.key_index(reduced.pop()) == full.result.best_index
    exact = {o[5] for o in out}
    assert exact == {(full.result.best_cost, full.result.best_index)}    # full doubles, same index
    out.sort()
    assert out[0][1] == 0 and out[0][2] == out[1][1] and out[1][2] == full.result.n_samples
    if which == "tie":
        assert full.result.best_index == 1          # the later of two equal minima, on rank 1


def test_exact_words_beat_the_40_bit_key_on_near_ties():
    """Two costs 1e-12 apart on different ranks: the 8-byte key cannot tell them apart (it resolves
    to the higher index), the slot words do (the reference compares full doubles,
    local_planner.cpp:460)."""
    a, b = 1.0, 1.0 + 1e-12
    ka, kb = sharding.pack_key(a, 3), sharding.pack_key(b, 9)
    assert sharding.key_index(min(ka, kb)) == 9            # documented limit of the packed key
    slots = list(sharding.winner_words(a, 3)) + list(sharding.winner_words(b, 9))
    assert sharding.reduce_words(slots) == (a, 3)
    # exact tie -> higher index; "none" ranks are ignored; costs above the reference's cap are none
    slots = list(sharding.winner_words(a, 3)) + list(sharding.winner_words(a, 9)) + list(sharding.winner_words(-1.0, -1))
    assert sharding.reduce_words(slots) == (a, 9)
    assert sharding.winner_words(1e7, 2) == (sharding.INT64_MAX, sharding.INT64_MAX)
    assert sharding.pack_key(1e7, 2) == K.KEY_NONE
    assert sharding.reduce_words([sharding.INT64_MAX] * 4) == (-1.0, -1)
