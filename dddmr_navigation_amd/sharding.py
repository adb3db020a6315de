"""Multi-GPU plumbing of the rollout tick.

Trajectories are independent given (robot pose, cloud, prune plan), so the
global sample list is cut into contiguous index ranges, one per rank (contiguity
keeps the reference's x-major order, which the last-wins tie-break needs), every
rank holds a full replica of the cloud, and the only exchange step is ONE 8-byte
min all-reduce of the packed argmin key (RCCL over xGMI with backend "nccl",
gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple

from . import _capi as K


def shard_range(rank: int, world_size: int, n: int) -> Tuple[int, int]:
    """[begin, end) of `rank`; identical to the library's split in dddmr_rollout_tick."""
    world_size = max(1, world_size)
    return (rank * n) // world_size, ((rank + 1) * n) // world_size


def pack_key(cost: float, global_index: int) -> int:
    return int(K.load_library().dddmr_rollout_pack_key(float(cost), int(global_index)))


def key_index(key: int) -> int:
    return int(K.load_library().dddmr_rollout_key_index(int(key)))


def all_reduce_key(key: int, device=None) -> int:
    """One min all-reduce of the 8-byte key over the default process group."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(key)
    t = torch.tensor([int(key)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())
