"""Time cough_spectrogram (the STFT stage alone) at B clips.  Usage: python tools/bench_stft.py [--batch 4096] [--launches 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from cough_detector_amd.hostcpu import bound_torch_threads

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--launches", type=int, default=30)
ap.add_argument("--magnitude", action="store_true")
ap.add_argument("--full-window", action="store_true")
args = ap.parse_args()
bound_torch_threads()
dev = torch.device("cuda:0")
pre = cda.AudioPreprocessor(device="cuda", use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                            use_spectral_contrast=False)
wav = torch.from_numpy(synth.make_clips(0, 256, peak_normalize=False)).to(dev).repeat(args.batch // 256, 1).contiguous()
spec = torch.empty((args.batch, 257, 101), dtype=torch.float32, device=dev)
kw = dict(power=1.0 if args.magnitude else 2.0, full_window=args.full_window, out=spec)
for _ in range(3):
    pre.spectrogram_batch(wav, **kw)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.launches):
    pre.spectrogram_batch(wav, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.launches
nbytes = args.batch * (64000 + 103828)
print(f"stft B={args.batch}: {ms:.4f} ms/launch  {nbytes / ms / 1e6:.1f} GB/s algorithmic  ({nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s)")
