"""Host-side phase times of the captured streaming tick (staging copy / graph launch / wait).  Diagnostic."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from cough_detector_amd.hostcpu import bound_torch_threads
from cough_detector_amd.streaming import MultiStreamDetector

bound_torch_threads(4)
S = 64
model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16_approx")
model.load_state_dict(synth.random_state_dict(seed=3))
det = MultiStreamDetector(model, S, confidence_threshold=0.7, clock=lambda: 0.0)
audio = torch.from_numpy(np.stack([synth.make_stream(100 + s, 6.0) for s in range(S)])).pin_memory()
for i in range(0, 20 * 1600, 1600):
    det.push(audio[:, i:i + 1600])
g = det._g
T = {"copy": [], "meta": [], "replay": [], "wait": [], "gpu": []}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(200):
    ch = audio[:, (rep % 30) * 1600:(rep % 30) * 1600 + 1600]
    g["done"].synchronize()
    t0 = time.perf_counter()
    g["h_chunks"].copy_(ch)
    t1 = time.perf_counter()
    meta = g["h_meta"].numpy()
    meta[:S] = det.written
    meta[S:] = det.next_start
    t2 = time.perf_counter()
    e0.record()
    g["full"].replay()
    e1.record()
    g["done"].record()
    t3 = time.perf_counter()
    while not g["done"].query():
        pass
    t4 = time.perf_counter()
    T["copy"].append(t1 - t0); T["meta"].append(t2 - t1); T["replay"].append(t3 - t2); T["wait"].append(t4 - t3)
    T["gpu"].append(e0.elapsed_time(e1) * 1e-3)
for k, v in T.items():
    print(f"{k:7s} p50 {np.percentile(v, 50) * 1e6:8.1f} us   p99 {np.percentile(v, 99) * 1e6:8.1f} us")
