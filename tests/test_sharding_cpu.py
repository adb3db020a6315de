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
        q.put((rank, b, e, key, red))
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
    out.sort()
    assert out[0][1] == 0 and out[0][2] == out[1][1] and out[1][2] == full.result.n_samples
    if which == "tie":
        assert full.result.best_index == 1          # the later of two equal minima, on rank 1
