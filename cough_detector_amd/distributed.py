"""Multi-GPU sharding of the clip stream: one process per GPU, RCCL over xGMI.

The reference is single-process; every 1 s window is independent end to end (all
reductions are per clip, BatchNorm is frozen), so the stream shards with NO data-path collective.
The only exchange publishes results: an all-gather of the (n_local, 2) logits, 8 B per clip.

Partitioning is round-robin: global clip ``i`` lives on rank ``i % W`` at local index ``i // W``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def local_count(n_total: int, rank: int, world: int) -> int:
    """How many of ``n_total`` round-robin-sharded clips rank ``rank`` owns."""
    return (n_total - rank + world - 1) // world if n_total > rank else 0


def local_indices(n_total: int, rank: int, world: int) -> torch.Tensor:
    """Global clip indices owned by ``rank`` in local order."""
    return torch.arange(rank, max(n_total, rank), world)


def gather_logits_round_robin(local_logits: torch.Tensor, n_total: Optional[int] = None,
                              out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """All-gather per-rank logits and un-interleave them into global clip order.

    ``local_logits``: (n_local, C) on this rank's device (n_local = local_count(n_total, rank, W);
    ranks may differ by one clip when W does not divide n_total).  Returns (n_total, C) on every rank:
    ``out[i] = logits_of_rank[i % W][i // W]``.  One collective; the rank-major -> clip-major
    permutation is a strided device copy.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local, c = local_logits.shape
    if n_total is None:
        n_total = n_local * world
    n_max = (n_total + world - 1) // world
    if n_local != local_count(n_total, rank, world):
        raise ValueError(f"rank {rank}: expected {local_count(n_total, rank, world)} local rows, got {n_local}")
    send = local_logits.contiguous()
    if n_local < n_max:                                   # ragged tail: pad to the common length
        send = torch.cat([send, send.new_zeros((n_max - n_local, c))], dim=0)
    stacked = torch.empty((world, n_max, c), dtype=send.dtype, device=send.device)
    if dist.get_backend(group) == "gloo":                 # CPU rehearsal path used by the tests
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
        stacked = torch.stack(parts, dim=0)
    else:
        dist.all_gather_into_tensor(stacked, send, group=group)
    full = stacked.permute(1, 0, 2).reshape(n_max * world, c)[:n_total]
    if out is None:
        return full.contiguous()
    if out.shape != (n_total, c):
        raise ValueError(f"out must be ({n_total}, {c})")
    out.copy_(full)
    return out
