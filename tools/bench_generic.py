"""Throughput of the generic-geometry featuriser chain (csrc/featurize_generic.hip) next to the tuned kernel.
Run on the GPU box: python tools/bench_generic.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cough_detector_amd as cda
from cough_detector_amd.hostcpu import bound_torch_threads

bound_torch_threads()
SHIPPED = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)
CASES = [("shipped geometry (tuned one-launch kernel)", dict(), 4096),
         ("f_max = 8000 Hz (dense bands up to bin 256)", dict(f_max=8000.0), 4096),
         ("80 mel / 20 MFCC / f_max 8000", dict(n_mels=80, n_mfcc=20, f_max=8000.0), 4096),
         ("2 s windows (201 frames)", dict(segment_duration=2.0), 2048),
         ("0.5 s windows (51 frames)", dict(segment_duration=0.5), 8192),
         ("22.05 kHz, hop 220, win 441", dict(sample_rate=22050, f_max=8000.0, hop_length=220, win_length=441), 4096),
         ("hop 200 (81 frames)", dict(hop_length=200), 4096),
         ("hop 128 / win 512 (126 frames)", dict(hop_length=128, win_length=512), 4096),
         ("hop 100 (161 frames)", dict(hop_length=100), 4096),
         ("hop 300 (54 frames)", dict(hop_length=300), 4096),
         ("63 mel bands", dict(n_mels=63), 4096),
         ("128 mel / 40 MFCC / f_max 8000", dict(n_mels=128, n_mfcc=40, f_max=8000.0), 4096),
         ("42 MFCCs (generic chain)", dict(n_mfcc=42), 4096),
         ("n_fft 1024, win 400 (radix-4 Stockham kernel)", dict(n_fft=1024), 4096),
         ("n_fft 256, win 256, hop 128", dict(n_fft=256, win_length=256, hop_length=128), 4096),
         ("n_fft 400 = win (DFT on the f32 matrix cores)", dict(n_fft=400), 4096),
         ("2 s + pre-emphasis", dict(segment_duration=2.0, use_pre_emphasis=True), 2048),
         ("2 s + PCEN", dict(segment_duration=2.0, use_pcen=True), 2048),
         ("2 s + delta-delta", dict(segment_duration=2.0, use_delta_delta=True), 2048),
         ("0.5 s + PCEN", dict(segment_duration=0.5, use_pcen=True), 8192),
         ("2 s, all constructor defaults but 4 contrast bands", dict(segment_duration=2.0, use_pcen=True, use_pre_emphasis=True,
                                                                    use_delta_delta=True, use_spectral_contrast=True,
                                                                    n_contrast_bands=4), 2048)]
only = sys.argv[1] if len(sys.argv) > 1 else ""      # substring filter, e.g. for a rocprofv3 run of one case
for name, kw, b in CASES:
    if only not in name:
        continue
    flags = {**SHIPPED, **{k: v for k, v in kw.items() if k.startswith("use_") or k == "n_contrast_bands"}}
    geom = {k: v for k, v in kw.items() if k not in flags}
    pre = cda.AudioPreprocessor(device="cuda", **geom, **flags)
    w = [torch.randn(b, pre.segment_samples, device="cuda") * 0.1 for _ in range(2)]
    out = torch.empty((b, pre.get_num_features(), pre._frames(pre.segment_samples)), device="cuda")
    for i in range(5):
        pre.featurize_batch(w[i % 2], normalize=True, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        pre.featurize_batch(w[i % 2], normalize=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    sec = b * pre.segment_samples / pre.sample_rate
    print(f"{name:55s} [{pre.kernel_path():14s}] B {b:5d}  {tuple(out.shape[1:])}  {ms:8.3f} ms  {b / ms / 1e3:7.2f} M clips/s  "
          f"{sec / ms * 1e3 / 3600:9.1f} audio-hours/s", flush=True)
