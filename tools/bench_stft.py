"""Time cough_spectrogram (the STFT stage alone) at B clips the way bench.py's `roofline_stft` does: device-generated
clips, 3 distinct batches in rotation (> 2 x the Infinity Cache), pre-warmed clocks, HIP events on the launch stream.
Usage: python tools/bench_stft.py [--batch 4096] [--launches 300] [--check]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from cough_detector_amd.hostcpu import bound_torch_threads

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--launches", type=int, default=300)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--magnitude", action="store_true")
ap.add_argument("--full-window", action="store_true")
ap.add_argument("--check", action="store_true", help="compare 64 clips with the CPU oracle first")
ap.add_argument("--prewarm-s", type=float, default=0.6, help="untimed pre-warm (0 under rocprofv3 --pmc)")
args = ap.parse_args()
bound_torch_threads()
dev = torch.device("cuda:0")
pre = cda.AudioPreprocessor(device="cuda", use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                            use_spectral_contrast=False)
B = args.batch
pool = torch.empty((3 * B, 16000), dtype=torch.float32, device=dev)
for r in range(3):
    synth.device_clips(r * B, B, out=pool[r * B:(r + 1) * B])
batches = [pool[r * B:(r + 1) * B] for r in range(3)]
spec = torch.empty((B, 257, 101), dtype=torch.float32, device=dev)
kw = dict(power=1.0 if args.magnitude else 2.0, full_window=args.full_window, out=spec)
if args.check:
    from oracle import featurizer as ofeat
    w = batches[0][:64].cpu()
    got = pre.spectrogram_batch(batches[0][:64], power=kw["power"], full_window=args.full_window).cpu()
    ref = ofeat.stft_power(w, win=512 if args.full_window else 400, power=kw["power"])
    err = ((got - ref).abs() / ref.abs().clamp_min(ref.abs().amax(dim=(1, 2), keepdim=True) * 1e-5)).max().item()
    print(f"check vs oracle (64 clips): max rel err {err:.2e}")
t0 = time.perf_counter()
while time.perf_counter() - t0 < args.prewarm_s:
    for i in range(20):
        pre.spectrogram_batch(batches[i % 3], **kw)
    torch.cuda.synchronize()
nbytes = B * (64000 + 103828)
for _ in range(args.rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.launches):
        pre.spectrogram_batch(batches[i % 3], **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.launches
    print(f"stft B={B} {os.environ.get('COUGH_AMD_LIB', 'lib').split('/')[-1]}: {ms:.4f} ms/launch  "
          f"{nbytes / ms / 1e6:.1f} GB/s algorithmic  ({nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
