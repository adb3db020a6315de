"""Multi-GPU plumbing of the rollout tick.

Trajectories are independent given (robot pose, cloud, prune plan), so the
global sample list is cut into contiguous index ranges, one per rank (contiguity
keeps the reference's x-major order, which the last-wins tie-break needs), every
rank holds a full replica of the cloud, and the only exchange step is ONE small
min all-reduce (RCCL over xGMI with backend "nccl", gloo in the CPU tests):
either of the 8-byte packed key, or -- exact for every pair of costs -- of a
slot vector holding (cost bits, -index) per rank (16 bytes per rank).
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


INT64_MAX = (1 << 63) - 1


def winner_words(cost: float, global_index: int) -> Tuple[int, int]:
    """(bit pattern of an acceptable cost, -index); (INT64_MAX, INT64_MAX) for "none" --
    the same words dddmr_rollout_winner_words produces from a tick result."""
    import struct

    if global_index < 0 or not (0.0 <= cost <= 9999999.0):
        return INT64_MAX, INT64_MAX
    return struct.unpack("<q", struct.pack("<d", float(cost)))[0], -int(global_index)


def reduce_words(slots) -> Tuple[float, int]:
    """Lexicographic minimum over the ranks' (cost bits, -index) words -> (cost, index);
    (-1.0, -1) when no rank has a winner.  Same rule as dddmr_rollout_resolve_words."""
    import struct

    best = (INT64_MAX, INT64_MAX)
    for r in range(len(slots) // 2):
        w = (int(slots[2 * r]), int(slots[2 * r + 1]))
        if w[0] != INT64_MAX and w < best:
            best = w
    if best[0] == INT64_MAX:
        return -1.0, -1
    return struct.unpack("<d", struct.pack("<q", best[0]))[0], -best[1]


def all_reduce_words(words: Tuple[int, int], rank: int, world_size: int, device=None):
    """ONE min all-reduce of the 2*world_size int64 slot vector; returns it as a list."""
    import torch
    import torch.distributed as dist

    t = torch.full((2 * max(1, world_size),), INT64_MAX, dtype=torch.int64,
                   device=device if device is not None else "cpu")
    t[2 * rank] = int(words[0])
    t[2 * rank + 1] = int(words[1])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return [int(v) for v in t.tolist()]
