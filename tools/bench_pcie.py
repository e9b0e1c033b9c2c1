"""PCIe-inclusive rate: the batch starts in pinned host memory each step (H2D copy + fused pipeline).
Reported in DESIGN.md only -- never as bench.py's `value`.  Usage: python tools/bench_pcie.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from cough_detector_amd.hostcpu import bound_torch_threads

bound_torch_threads()
B = 4096
dev = torch.device("cuda:0")
pre = cda.AudioPreprocessor(device="cuda", use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                            use_spectral_contrast=False)
model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16_approx")
model.load_state_dict(synth.random_state_dict(seed=3))
model.to(dev).eval()
pipe = cda.CoughPipeline(pre, model)
host = torch.from_numpy(synth.make_clips(0, 256, peak_normalize=False)).repeat(B // 256, 1).pin_memory()
bufs = [torch.empty((B, 16000), dtype=torch.float32, device=dev) for _ in range(2)]
copy_s, comp_s = torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
for mode in ("serial", "double-buffered"):
    for it in range(3):
        bufs[0].copy_(host, non_blocking=True)
        pipe(bufs[0], normalize=True)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    if mode == "serial":
        for it in range(K):
            bufs[0].copy_(host, non_blocking=True)
            pipe(bufs[0], normalize=True)
    else:
        ev_copied = [torch.cuda.Event() for _ in range(2)]
        ev_used = [torch.cuda.Event() for _ in range(2)]
        for e in ev_used:
            e.record(comp_s)
        for it in range(K):
            k = it & 1
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(ev_used[k])
                bufs[k].copy_(host, non_blocking=True)
                ev_copied[k].record(copy_s)
            comp_s.wait_event(ev_copied[k])
            pipe(bufs[k], normalize=True)
            ev_used[k].record(comp_s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{mode:16s}: {dt * 1e3:.3f} ms/step  {B / dt / 1e6:.3f} M clips/s  ({B * 64000 / dt / 1e9:.1f} GB/s host->device)")
