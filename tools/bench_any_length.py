import os, sys, torch
sys.path.insert(0, os.getcwd())
import cough_detector_amd as cda
SH = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)
pre = cda.AudioPreprocessor(device="cuda", **SH)
for n, b in ((16000, 4096), (12345, 4096), (24000, 2731), (40000, 1638)):
    w = torch.randn(b, n, device="cuda") * 0.1
    for _ in range(5): pre.featurize_batch(w, normalize=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): pre.featurize_batch(w, normalize=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"shipped handle, waveforms of {n:6d} samples, B {b}: {ms:.3f} ms  {b * n / 16000 / ms * 1e3 / 1e6:.2f} M audio-seconds/s", flush=True)
