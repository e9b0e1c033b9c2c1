#!/usr/bin/env python3
"""Streaming latency benchmark (BASELINE.json configs[4]): S concurrent 16 kHz mic-like streams, 0.1 s
chunks, 1 s window / 0.25 s hop (256 windows/s at S = 64 in real time).  Measures, per tick that completes
at least one window, the wall time from "chunks handed to push()" to "probabilities on the host"
(chunk upload + ring write + window gather + featurise + classifier + download), and the sustained
windows/s when ticks are issued back to back.  One process = one GPU; with torch.distributed.run each
rank serves streams s = rank, rank + W, ... (no collective: per-stream state stays on its rank).

    python bench_streaming.py --streams 64 --seconds 20 [--stagger]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--dtype", default="bf16x3", choices=["bf16x3", "bf16_approx", "fp32"])
    ap.add_argument("--stagger", action="store_true", help="de-phase the streams so windows complete on every tick")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    from cough_detector_amd.hostcpu import bound_torch_threads
    bound_torch_threads(4)

    import cough_detector_amd as cda
    from cough_detector_amd import synth
    from cough_detector_amd.streaming import MultiStreamDetector

    my_streams = list(range(rank, args.streams, world))
    S = len(my_streams)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=args.dtype)
    model.load_state_dict(synth.random_state_dict(seed=3))
    now = {"t": 0.0}
    det = MultiStreamDetector(model, S, confidence_threshold=0.7, clock=lambda: now["t"])
    audio = np.stack([synth.make_stream(100 + s, args.seconds) for s in my_streams])
    chunk = 1600
    pinned = torch.from_numpy(audio).pin_memory()
    # warm-up (allocator, kernels)
    for i in range(0, 12 * chunk, chunk):
        det.push(pinned[:, i:i + chunk])
    det.reset()
    if args.stagger:       # stream k starts (k * 400) % 4000 samples early: window completions spread over ticks
        for k in range(S):
            lead = (k * 400) % 4000
            if lead:
                det.push(pinned[k:k + 1, :lead], stream_ids=[k])
    for p in det.window_probs:
        p.clear()
    torch.cuda.synchronize()

    lat, nwin, ticks = [], 0, 0
    t_all = time.perf_counter()
    for i in range(0, audio.shape[1] - chunk + 1, chunk):
        now["t"] = (i + chunk) / 16000.0
        before = sum(len(p) for p in det.window_probs)
        t0 = time.perf_counter()
        det.push(pinned[:, i:i + chunk])
        dt = time.perf_counter() - t0
        done = sum(len(p) for p in det.window_probs) - before
        ticks += 1
        if done:
            lat.append(dt * 1e3)
            nwin += done
    wall = time.perf_counter() - t_all
    lat = np.array(lat)
    print(json.dumps({
        "metric": "window->probability latency, streaming", "rank": rank, "n_gpus": world, "streams_this_gpu": S,
        "dtype": args.dtype, "chunk_s": 0.1, "window_s": 1.0, "hop_s": 0.25, "staggered": bool(args.stagger),
        "ticks": ticks, "ticks_with_windows": int(len(lat)), "windows": int(nwin),
        "latency_ms_p50": round(float(np.percentile(lat, 50)), 3), "latency_ms_p99": round(float(np.percentile(lat, 99)), 3),
        "latency_ms_max": round(float(lat.max()), 3), "worst_ticks": [int(i) for i in np.argsort(lat)[-3:]],
        "sustained_windows_per_s": round(nwin / wall, 1),
        "real_time_need_windows_per_s": round(S * 4.0, 1),
        "stream_seconds_per_wall_second": round(args.seconds / wall, 1)}), flush=True)


if __name__ == "__main__":
    main()
