#!/usr/bin/env python3
"""Streaming latency benchmark (BASELINE.json configs[4]): S concurrent 16 kHz mic-like streams, 0.1 s
chunks, 1 s window / 0.25 s hop (256 windows/s at S = 64 in real time).  Measures, per tick that completes
at least one window, the wall time from "chunks handed to push()" to "probabilities on the host"
(chunk upload + ring write + window gather + featurise + classifier + download), and the sustained
windows/s when ticks are issued back to back.

One process = one GPU.  ``--gpus N`` (N > 1) without a torchrun environment makes this process the PARENT: it
starts ``python -m torch.distributed.run --nproc-per-node N ... bench_streaming.py ...`` as a child (never an
exec, never a GPU call of its own) and relays the ONE JSON line rank 0 prints.  Under torchrun each rank serves the
streams s = rank, rank + W, ... -- no data-path collective: a stream's ring buffer and smoothing state stay on its
rank -- and after the run ONE small all-gather collects every rank's latency samples and record, so that rank 0
reports the global p50 / p99 / max over all streams, the per-rank records and the aggregate windows/s.

    python bench_streaming.py --streams 64 --seconds 20 [--stagger] [--gpus 8]
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks of one node (one per GPU); > 1 self-launches under torchrun")
    ap.add_argument("--streams", type=int, default=64, help="concurrent streams over ALL ranks")
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--dtype", default="bf16x3", choices=["bf16x3", "bf16_approx", "fp32"])
    ap.add_argument("--stagger", action="store_true", help="de-phase the streams so windows complete on every tick")
    return ap.parse_args(argv)


def build_launch_cmd(argv, n_ranks: int, port: int):
    """The child command a ``--gpus N`` parent runs: torchrun's module entry with one rank per GPU of this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(args, argv) -> int:
    """Parent of a multi-rank run: subprocess only, never touches the GPU runtime; relays rank 0's single JSON line."""
    from cough_detector_amd.hostcpu import explicit_device_limit
    limit = explicit_device_limit()
    if limit is not None and limit < args.gpus:
        print(f"bench_streaming.py --gpus {args.gpus}: only {limit} device(s) visible on this node "
              f"(ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES names {limit})", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    proc = subprocess.run(build_launch_cmd(argv, args.gpus, port), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    for ln in proc.stdout.splitlines():
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if proc.returncode != 0:
        return proc.returncode
    return 0 if lines else 1


def aggregate(rank_latencies, rank_records, args, backend: str, world: int) -> dict:
    """ONE line for the whole job from every rank's latency samples (ms, one per tick that completed a window) and record
    ``{rank, device, streams, ticks, ticks_with_windows, windows, wall_s, ...}``: percentiles over ALL samples of all ranks (a
    stream's window is late when ITS rank's tick is late), the job's windows/s = all windows / the slowest rank's wall time."""
    lat = np.concatenate([np.asarray(l, dtype=np.float64) for l in rank_latencies]) if rank_latencies else np.zeros(0)
    windows = int(sum(r["windows"] for r in rank_records))
    wall = max(float(r["wall_s"]) for r in rank_records)
    streams = int(sum(r["streams"] for r in rank_records))

    def pct(q):
        return round(float(np.percentile(lat, q)), 3) if lat.size else None

    per_rank = []
    for r, l in zip(rank_records, rank_latencies):
        l = np.asarray(l, dtype=np.float64)
        per_rank.append({**{k: (int(v) if k != "wall_s" else round(float(v), 4)) for k, v in r.items()},
                         "latency_ms_p50": round(float(np.percentile(l, 50)), 3) if l.size else None,
                         "latency_ms_p99": round(float(np.percentile(l, 99)), 3) if l.size else None,
                         "latency_ms_max": round(float(l.max()), 3) if l.size else None})
    return {
        "metric": "window->probability latency p50, streaming (configs[4])", "value": pct(50), "unit": "ms",
        "higher_is_better": False, "n_gpus": world, "scaling": "weak (streams sharded by stream id, no data-path collective)",
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"configs[4]: {streams} concurrent 16 kHz streams, 0.1 s chunks, 1 s window / 0.25 s hop, "
                               "window complete -> probability on the host",
                   "streams": streams, "chunk_s": 0.1, "window_s": 1.0, "hop_s": 0.25, "staggered": bool(args.stagger),
                   "stream_seconds": args.seconds, "sharding": f"stream s -> rank s mod {world}"},
        "latency_ms_p50": pct(50), "latency_ms_p99": pct(99), "latency_ms_max": round(float(lat.max()), 3) if lat.size else None,
        "ticks_with_windows": int(lat.size), "windows": windows,
        "sustained_windows_per_s": round(windows / wall, 1) if wall > 0 else None,
        "real_time_need_windows_per_s": round(streams * 4.0, 1),
        "stream_seconds_per_wall_second": round(args.seconds / wall, 1) if wall > 0 else None,
        "rccl_world": world, "backend": backend, "ranks": per_rank}


def gather_latencies(lat, record: dict, device):
    """Every rank's latency samples and record on every rank: one all-gather of the sample counts' maximum-padded vectors (NaN
    padding) plus ``distributed.gather_rank_records``.  World 1 without a process group: this rank's own."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [list(lat)], [record]
    from cough_detector_amd.distributed import gather_rank_records
    world = dist.get_world_size()
    records = gather_rank_records(record, device)
    n_max = max(int(r["ticks_with_windows"]) for r in records)
    mine = torch.full((max(n_max, 1),), float("nan"), dtype=torch.float64, device=device)
    if len(lat):
        mine[:len(lat)] = torch.as_tensor(np.asarray(lat, dtype=np.float64), device=device)
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        rows = torch.stack(parts).cpu()
    else:
        rows = torch.empty((world, mine.numel()), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(rows, mine)
        rows = rows.cpu()
    return [[float(v) for v in rows[r] if v == v] for r in range(world)], records


def run_rank(args) -> int:
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = os.environ.get("COUGH_BENCH_SHARE_GPU") == "1"      # rehearsal on a one-GPU box: every rank on device 0, gloo
    backend = os.environ.get("COUGH_BENCH_BACKEND", "nccl")
    dist = None
    if world > 1 and os.environ.get("COUGH_BENCH_NUMA", "1") == "1":
        from cough_detector_amd.hostcpu import bind_to_gpu_numa
        bind_to_gpu_numa(0 if share_gpu else local_rank)      # before anything touches the GPU
    device_index = 0 if share_gpu else local_rank
    if device_index >= torch.cuda.device_count():
        print(f"bench_streaming.py: rank {rank}: no device {device_index} (torch sees {torch.cuda.device_count()})", file=sys.stderr)
        return 2
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    saved_stdout = None
    if "WORLD_SIZE" in os.environ:
        import torch.distributed as dist
        saved_stdout = os.dup(1)
        os.dup2(2, 1)                                   # RCCL's banner goes to stderr: rank 0 prints exactly one JSON line
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        dist.barrier()
    from cough_detector_amd.hostcpu import bound_torch_threads
    bound_torch_threads(4)

    import cough_detector_amd as cda
    from cough_detector_amd import synth
    from cough_detector_amd.streaming import MultiStreamDetector

    my_streams = list(range(rank, args.streams, world))
    S = len(my_streams)
    lat, nwin, ticks, wall = [], 0, 0, 0.0
    if S:
        model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=args.dtype)
        model.load_state_dict(synth.random_state_dict(seed=3))
        now = {"t": 0.0}
        det = MultiStreamDetector(model, S, confidence_threshold=0.7, clock=lambda: now["t"])
        audio = np.stack([synth.make_stream(100 + s, args.seconds) for s in my_streams])
        chunk = 1600
        pinned = torch.from_numpy(audio).pin_memory()
        for i in range(0, 12 * chunk, chunk):           # warm-up (allocator, kernels, graph capture)
            det.push(pinned[:, i:i + chunk])
        det.reset()
        if args.stagger:       # stream k starts (k * 400) % 4000 samples early: window completions spread over ticks
            for k in range(S):
                lead = (k * 400) % 4000
                if lead:
                    det.push(pinned[k:k + 1, :lead], stream_ids=[k])
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t_all = time.perf_counter()
        for i in range(0, audio.shape[1] - chunk + 1, chunk):
            now["t"] = (i + chunk) / 16000.0
            before = det.windows_seen
            t0 = time.perf_counter()
            det.push(pinned[:, i:i + chunk])
            dt = time.perf_counter() - t0
            done = det.windows_seen - before
            ticks += 1
            if done:
                lat.append(dt * 1e3)
                nwin += done
        wall = time.perf_counter() - t_all
    elif dist:
        dist.barrier()
    record = {"rank": rank, "device": device_index, "streams": S, "ticks": ticks, "ticks_with_windows": len(lat),
              "windows": nwin, "wall_s": wall}
    all_lat, records = gather_latencies(lat, record, dev)
    if rank == 0:
        line = aggregate(all_lat, records, args, ("rccl (torch.distributed 'nccl' on ROCm)" if backend == "nccl" else backend)
                         if dist else "none (single process)", world)
        if share_gpu or (dist and backend != "nccl"):
            line["config"]["sharding"] += f" -- REHEARSAL: backend {backend}, ranks share one GPU: not a scaling measurement"
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)
    if dist:
        dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    force_dist = os.environ.get("COUGH_BENCH_FORCE_DIST") == "1"    # rehearse the multi-rank path with one rank
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
