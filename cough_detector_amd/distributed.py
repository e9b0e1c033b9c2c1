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


class BucketedLogitsGather:
    """Publishes the logits of a round-robin-sharded clip STREAM with few, larger collectives.

    Every step a rank scores ``batch`` consecutive local clips (the last steps of a ragged stream fewer); ``push``
    copies the step's logits into a send bucket and, every ``every`` steps, ONE all-gather exchanges the bucket
    (``every * batch * 8`` bytes per rank) on RCCL's stream while the next bucket is being computed.  The steps of a
    bucket cover a contiguous range of global clips starting at a multiple of the world size, so the bucket is
    round-robin-sharded exactly like a single large batch and ``gather_logits_start / _finish`` apply unchanged.
    Per-step exchanges (``every = 1``) cost a collective launch and an un-interleave copy per 0.7 ms step; the
    bucket amortises both (xGMI ring latency, not bandwidth, prices an 8-byte-per-clip exchange).

    ``out`` holds the most recently finished bucket in global clip order; ``drain()`` finishes everything in flight.
    """

    def __init__(self, batch: int, every: int, device, classes: int = 2, dtype=torch.float32, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.every = max(1, int(every))
        rows = self.every * batch
        self.send = [torch.empty((rows, classes), dtype=dtype, device=device) for _ in range(2)]
        self.out = torch.empty((rows * self.world, classes), dtype=dtype, device=device)
        self.cur, self.rows, self.n_total, self.steps = 0, 0, 0, 0
        self.inflight: Optional[_Gather] = None
        self.last_n = 0
        self.collectives = 0        # exchanges started
        self.finished = 0           # exchanges whose result has been written to ``out``

    def push(self, logits: torch.Tensor, n_total_step: int) -> None:
        """``logits``: this rank's rows of one step; ``n_total_step``: the step's clips over all ranks."""
        n = logits.shape[0]
        if self.rows + n > self.send[self.cur].shape[0]:
            raise ValueError("step larger than the batch the bucket was sized for")
        if n:
            self.send[self.cur][self.rows:self.rows + n].copy_(logits)
        self.rows += n
        self.n_total += n_total_step
        self.steps += 1
        if self.steps == self.every:
            self.flush()

    def flush(self) -> None:
        """Start the exchange of a (possibly partial) bucket; at most one exchange is in flight."""
        if self.steps == 0:
            return
        self._finish()
        self.inflight = gather_logits_start(self.send[self.cur][:self.rows], n_total=self.n_total, group=self.group)
        self.collectives += 1
        self.cur ^= 1
        self.rows, self.n_total, self.steps = 0, 0, 0

    def _finish(self) -> None:
        if self.inflight is not None:
            h, self.inflight = self.inflight, None
            gather_logits_finish(h, out=self.out[:h.n_total])
            self.last_n = h.n_total
            self.finished += 1

    def drain(self) -> torch.Tensor:
        self.flush()
        self._finish()
        return self.out[:self.last_n]


def stream_shard(total_clips: int, batch: int, rank: int, world: int, device):
    """configs[3]'s input: this rank's round-robin shard of a ``total_clips``-clip synthetic stream, generated ON THE
    DEVICE from the clips' global indices (clip ``g`` = ``synth.make_clip_counter(g)``, the recipe of
    ``/root/reference/setup_coughvid.py:381-441``) and kept resident in HBM (64 KB per clip).

    Returns ``(batches, step_total)``: ``batches[j]`` holds local clips ``j*batch .. (j+1)*batch-1`` of this rank
    (global clips ``rank + (j*batch + k)*world``), possibly short or empty on the last steps of a ragged stream;
    ``step_total[j]`` is the number of clips ALL ranks score in step ``j``.  Every rank runs the same number of
    steps (those of the fullest rank), so the bucketed exchange below stays collective."""
    from . import synth
    n_local = local_count(total_clips, rank, world)
    steps = (local_count(total_clips, 0, world) + batch - 1) // batch
    pool = torch.empty((max(n_local, 1), synth.N), dtype=torch.float32, device=device)
    for j in range(0, n_local, batch):
        c = min(batch, n_local - j)
        synth.device_clips(rank + j * world, c, seed_stride=world, out=pool[j:j + c])
    batches = [pool[j * batch:min((j + 1) * batch, n_local)] for j in range(steps)]
    step_total = [min(total_clips, (j + 1) * batch * world) - j * batch * world for j in range(steps)]
    return batches, step_total


def score_stream(pipeline, total_clips: int, batch: int = 4096, every: int = 8, group=None,
                 rank: Optional[int] = None, world: Optional[int] = None, device=None, shard=None) -> torch.Tensor:
    """Score a ``total_clips``-clip synthetic stream sharded round-robin over the ranks of ``group`` and return the
    logits of EVERY clip, in global clip order, on every rank: ``(total_clips, 2)``.

    This is configs[3] end to end: ``stream_shard`` -> ``CoughPipeline`` per step -> ``BucketedLogitsGather`` (one
    RCCL all-gather per ``every`` steps, overlapped with the next bucket) -> un-interleave.  Without an initialised
    process group (``rank`` / ``world`` given explicitly, or a single process) it returns only this rank's rows in
    local order -- the gather needs its peers.  ``shard`` replaces ``stream_shard`` (same signature; the gloo tests
    feed index-carrying CPU batches through a stand-in pipeline)."""
    have_group = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank(group) if have_group else 0
    if world is None:
        world = dist.get_world_size(group) if have_group else 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    batches, step_total = (shard or stream_shard)(total_clips, batch, rank, world, device)
    if not have_group:
        parts = [pipeline(b, normalize=True) for b in batches if b.shape[0]]
        return torch.cat(parts) if parts else torch.empty((0, 2), device=device)
    pub = BucketedLogitsGather(batch, every, device, group=group)
    full = torch.empty((total_clips, 2), dtype=torch.float32, device=device)
    done, copied = 0, 0

    def collect():
        nonlocal done, copied
        if pub.finished > copied:                    # a bucket landed in ``out``: global clips done .. done+last_n-1
            full[done:done + pub.last_n].copy_(pub.out[:pub.last_n])
            done += pub.last_n
            copied = pub.finished

    for j, wav in enumerate(batches):
        pub.push(pipeline(wav, normalize=True), step_total[j])
        collect()
    pub.flush()
    collect()
    pub.drain()
    collect()
    if done != total_clips:
        raise RuntimeError(f"stream exchange incomplete: {done} of {total_clips} clips gathered")
    return full


def gather_rank_records(record: dict, device, group=None) -> list:
    """One small all-gather of a flat numeric record per rank (same keys on every rank); returns the list of the
    ranks' records in rank order, on every rank.  ``bench.py`` uses it after the timed region to put every rank's own
    device / clock / kernel times / exchange counters on the N > 1 line."""
    keys = sorted(record)
    mine = torch.tensor([float(record[k]) for k in keys], dtype=torch.float64, device=device)
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        rows = torch.stack(parts).cpu()
    else:
        rows = torch.empty((world, len(keys)), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(rows, mine, group=group)
        rows = rows.cpu()
    out = []
    for r in range(world):
        rec = {}
        for j, k in enumerate(keys):
            v = float(rows[r, j])
            rec[k] = int(v) if v == int(v) and not k.endswith("_ms") and k != "ms_per_step" else round(v, 4)
        out.append(rec)
    return out
