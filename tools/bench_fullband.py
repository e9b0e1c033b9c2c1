"""Featurise-only and waveform -> logits throughput of the full-band one-launch featuriser next to the shipped one
(VERDICT r04 item 2).  Run on the GPU box: python tools/bench_fullband.py [--iters 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cough_detector_amd as cda
from cough_detector_amd import synth
from cough_detector_amd.hostcpu import bound_torch_threads

bound_torch_threads()
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--batch", type=int, default=4096)
args = ap.parse_args()
SHIPPED = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)
CASES = [("shipped (f_max 4000, 64 mel, 13 MFCC)", dict()),
         ("f_max 8000, 64 mel, 13 MFCC", dict(f_max=8000.0)),
         ("f_max 8000, 80 mel, 20 MFCC", dict(n_mels=80, n_mfcc=20, f_max=8000.0)),
         ("f_max 8000, 128 mel, 13 MFCC", dict(n_mels=128, f_max=8000.0)),
         ("f_max 8000, 40 mel, 13 MFCC", dict(n_mels=40, f_max=8000.0)),
         ("f_max 8000, 64 mel + pre-emphasis + delta-delta", dict(f_max=8000.0, use_pre_emphasis=True, use_delta_delta=True))]
B = args.batch
w = [synth.device_clips(i * B, B) for i in range(3)]          # 750 MiB in rotation: reads come from HBM
sd = synth.random_state_dict(seed=3)
for name, kw in CASES:
    flags = {**SHIPPED, **{k: v for k, v in kw.items() if k.startswith("use_")}}
    geom = {k: v for k, v in kw.items() if k not in flags}
    pre = cda.AudioPreprocessor(device="cuda", **geom, **flags)
    out = torch.empty((B, pre.get_num_features(), 101), device="cuda")
    for i in range(10):
        pre.featurize_batch(w[i % 3], normalize=True, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.iters):
        pre.featurize_batch(w[i % 3], normalize=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    line = f"{name:50s} [{pre.kernel_path():14s}] featurise {ms:7.4f} ms = {B / ms / 1e3:6.2f} M clips/s"
    if pre.get_num_features() in (90, 103):
        model = cda.create_model("residual", n_mels=pre.get_num_features(), num_classes=2, in_channels=1, compute_dtype="bf16x3")
        model.load_state_dict(sd)
        model.eval()
        pipe = cda.CoughPipeline(pre, model)
        for i in range(10):
            pipe(w[i % 3], normalize=True)
        e0.record()
        for i in range(args.iters):
            pipe(w[i % 3], normalize=True)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / args.iters
        line += f" | waveform -> logits (bf16x3) {ms2:7.4f} ms = {B / ms2 / 1e3:5.2f} M clips/s"
    print(line, flush=True)
