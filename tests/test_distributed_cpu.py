"""N > 1 path on CPU: world_size-2 / 3 gloo processes exercise the round-robin shard + all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cough_detector_amd import distributed as cdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = cdist.local_indices(n_total, rank, world)
        assert len(idx) == cdist.local_count(n_total, rank, world)
        # the "logits" of global clip i are (i, -i): any mis-ordering is visible
        local = torch.stack([idx.float(), -idx.float()], dim=1)
        full = cdist.gather_logits_round_robin(local, n_total=n_total)
        want = torch.stack([torch.arange(n_total).float(), -torch.arange(n_total).float()], dim=1)
        assert torch.equal(full, want), (rank, full[:8])
        out = torch.empty(n_total, 2)
        assert cdist.gather_logits_round_robin(local, n_total=n_total, out=out) is out and torch.equal(out, want)
        # two exchanges in flight, finished out of step with their launch (how bench.py overlaps them with compute)
        h1 = cdist.gather_logits_start(local, n_total=n_total)
        h2 = cdist.gather_logits_start(local * 2.0, n_total=n_total)
        assert torch.equal(cdist.gather_logits_finish(h1), want)
        assert torch.equal(cdist.gather_logits_finish(h2, out=out), want * 2.0)
    finally:
        dist.destroy_process_group()


def _bucket_worker(rank, world, port, n_total, batch, every):
    """A sharded stream scored in steps of ``batch`` local clips, published through BucketedLogitsGather: every
    finished bucket must hold its global clips in order, ragged last bucket included."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = cdist.local_indices(n_total, rank, world)[:cdist.local_count(n_total, rank, world)]
        n_steps = (cdist.local_count(n_total, 0, world) + batch - 1) // batch       # steps of the fullest rank
        pub = cdist.BucketedLogitsGather(batch, every, device="cpu")
        seen = []

        def collect():
            if pub.finished > len(seen):             # a bucket was finished: ``out`` holds it in global clip order
                seen.append(pub.out[:pub.last_n].clone())

        for j in range(n_steps):
            mine = idx[j * batch:(j + 1) * batch].float()
            step_total = min(n_total, (j + 1) * batch * world) - j * batch * world
            pub.push(torch.stack([mine, -mine], dim=1), step_total)
            collect()
        pub.flush()
        collect()
        pub.drain()
        collect()
        got = torch.cat(seen)
        want = torch.stack([torch.arange(n_total).float(), -torch.arange(n_total).float()], dim=1)
        assert torch.equal(got, want), (rank, got.shape, want.shape)
        assert pub.collectives == (n_steps + every - 1) // every
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,batch,every", [(2, 100, 8, 3), (2, 64, 8, 4), (3, 50, 4, 2), (2, 37, 8, 1)])
def test_bucketed_stream_gather_gloo(world, n_total, batch, every):
    mp.spawn(_bucket_worker, args=(world, _free_port(), n_total, batch, every), nprocs=world, join=True)


@pytest.mark.parametrize("world,n_total", [(2, 64), (2, 37), (3, 10)])
def test_round_robin_gather_gloo(world, n_total):
    mp.spawn(_worker, args=(world, _free_port(), n_total), nprocs=world, join=True)


def test_shard_helpers():
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 9, 4096):
            counts = [cdist.local_count(n, r, world) for r in range(world)]
            assert sum(counts) == n and max(counts) - min(counts) <= 1
            allidx = torch.cat([cdist.local_indices(n, r, world) for r in range(world)]).sort().values
            assert torch.equal(allidx, torch.arange(n))


def _index_shard(total, batch, rank, world, device):
    """Stand-in for ``stream_shard`` on CPU: a batch row carries its clip's GLOBAL index instead of a waveform."""
    idx = cdist.local_indices(total, rank, world)[:cdist.local_count(total, rank, world)].float()
    steps = (cdist.local_count(total, 0, world) + batch - 1) // batch
    batches = [idx[j * batch:(j + 1) * batch, None] for j in range(steps)]
    step_total = [min(total, (j + 1) * batch * world) - j * batch * world for j in range(steps)]
    return batches, step_total


def _score_worker(rank, world, port, n_total, batch, every):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake_pipeline = lambda b, normalize=True: torch.cat([b, -b], dim=1)      # "logits" of clip i = (i, -i)
        full = cdist.score_stream(fake_pipeline, n_total, batch=batch, every=every, device="cpu", shard=_index_shard)
        want = torch.stack([torch.arange(n_total).float(), -torch.arange(n_total).float()], dim=1)
        assert torch.equal(full, want), (rank, full[:6], full[-6:])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,batch,every", [(2, 101, 8, 3), (3, 50, 4, 2), (2, 7, 8, 8), (2, 33, 16, 1)])
def test_score_stream_returns_every_clip_in_global_order_gloo(world, n_total, batch, every):
    """configs[3]'s driver loop (shard -> score -> bucketed exchange -> collect) with W > 1, ragged tails and a
    rank whose last step is empty."""
    mp.spawn(_score_worker, args=(world, _free_port(), n_total, batch, every), nprocs=world, join=True)


def test_stream_shard_bookkeeping_matches_score_stream_contract():
    for world in (1, 2, 3, 8):
        for total, batch in ((0, 4), (1, 4), (9, 4), (64, 8), (37, 8)):
            shards = [_index_shard(total, batch, r, world, "cpu") for r in range(world)]
            assert len({len(b) for b, _ in shards}) == 1                    # same number of steps on every rank
            for j, tot in enumerate(shards[0][1]):
                assert sum(s[0][j].shape[0] for s in shards) == tot
            assert sum(shards[0][1]) == total


def _records_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        recs = cdist.gather_rank_records({"rank": rank, "device": rank, "numa_node": -1, "ms_per_step": 0.65 + rank,
                                          "k1_ms": 0.27, "cls_ms": 0.38, "collectives_started": 3 + rank,
                                          "collectives_finished": 3 + rank}, "cpu")
        assert [r["rank"] for r in recs] == list(range(world)) and [r["device"] for r in recs] == list(range(world))
        assert all(isinstance(r["device"], int) and isinstance(r["ms_per_step"], float) for r in recs)
        assert recs[world - 1]["ms_per_step"] == pytest.approx(0.65 + world - 1)
        assert recs[1]["collectives_started"] == 4 and recs[0]["numa_node"] == -1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rank_records_are_gathered_in_rank_order_gloo(world):
    """bench.py's N > 1 line: every rank's own device / clock / kernel times / exchange counters."""
    mp.spawn(_records_worker, args=(world, _free_port()), nprocs=world, join=True)


def _latency_worker(rank, world, port):
    """bench_streaming.py's end-of-run exchange: ragged per-rank latency samples (a rank may have none) + records."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench_streaming as bs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = [0.1 * (rank + 1) + 0.01 * i for i in range(5 - 2 * rank)] if rank < 2 else []     # 5, 3, 0 samples
        rec = {"rank": rank, "device": 0, "streams": 22 - rank, "ticks": 50, "ticks_with_windows": len(mine),
               "windows": 10 * len(mine), "wall_s": 1.0 + rank}
        lats, recs = bs.gather_latencies(mine, rec, torch.device("cpu"))
        assert [len(l) for l in lats] == [5, 3, 0][:world] and [r["rank"] for r in recs] == list(range(world))
        assert lats[rank] == pytest.approx(mine) and recs[rank]["streams"] == 22 - rank
        line = bs.aggregate(lats, recs, bs.parse_args(["--streams", "64"]), "gloo", world)
        assert line["windows"] == sum(10 * n for n in [5, 3, 0][:world]) and len(line["ranks"]) == world
        assert line["sustained_windows_per_s"] == round(line["windows"] / float(world), 1)          # slowest rank: wall = world s
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_streaming_bench_gathers_ragged_latency_samples_gloo(world):
    mp.spawn(_latency_worker, args=(world, _free_port()), nprocs=world, join=True)
