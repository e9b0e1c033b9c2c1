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


class _Gather:
    """An all-gather in flight (``gather_logits_start``)."""
    __slots__ = ("work", "stacked", "parts", "n_total", "n_max", "world", "c")


def gather_logits_start(local_logits: torch.Tensor, n_total: Optional[int] = None, group=None) -> _Gather:
    """Launch the all-gather of this rank's logits and return without waiting: RCCL runs it on its own stream, so
    the caller's next kernels overlap the exchange (8 B per clip: pure latency, worth hiding)."""
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
    h = _Gather()
    h.n_total, h.n_max, h.world, h.c = n_total, n_max, world, c
    h.parts = None
    if dist.get_backend(group) == "gloo":                 # CPU rehearsal path used by the tests
        h.parts = [torch.empty_like(send) for _ in range(world)]
        h.stacked = None
        h.work = dist.all_gather(h.parts, send, group=group, async_op=True)
    else:
        h.stacked = torch.empty((world, n_max, c), dtype=send.dtype, device=send.device)
        h.work = dist.all_gather_into_tensor(h.stacked, send, group=group, async_op=True)
    return h


def gather_logits_finish(h: _Gather, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Wait for the exchange (a stream-side wait with RCCL, not a host sync) and un-interleave into global clip
    order: ``out[i] = logits_of_rank[i % W][i // W]`` (a strided device copy)."""
    h.work.wait()
    stacked = torch.stack(h.parts, dim=0) if h.parts is not None else h.stacked
    full = stacked.permute(1, 0, 2).reshape(h.n_max * h.world, h.c)[:h.n_total]
    if out is None:
        return full.contiguous()
    if out.shape != (h.n_total, h.c):
        raise ValueError(f"out must be ({h.n_total}, {h.c})")
    out.copy_(full)
    return out


def gather_logits_round_robin(local_logits: torch.Tensor, n_total: Optional[int] = None,
                              out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """All-gather per-rank logits and un-interleave them into global clip order.

    ``local_logits``: (n_local, C) on this rank's device (n_local = local_count(n_total, rank, W);
    ranks may differ by one clip when W does not divide n_total).  Returns (n_total, C) on every rank.
    One collective; the rank-major -> clip-major permutation is a strided device copy."""
    return gather_logits_finish(gather_logits_start(local_logits, n_total, group), out)
