"""Forward time of the three classifiers from resident feature images.  Usage: python tools/bench_models.py [--batch 4096]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cough_detector_amd as cda
from cough_detector_amd.hostcpu import bound_torch_threads

MACS = {"residual": 21432800, "standard": None, "small": None}


def macs(kind, h=90, w=101):
    if kind == "residual":
        return 21432800
    if kind == "standard":
        tot, c = 0, 1
        for n in (32, 64, 128, 256):
            tot += h * w * 9 * c * n
            h, w, c = h // 2, w // 2, n
        return tot + 256 * 128 + 128 * 2
    tot = h * w * 9 * 16
    h, w = h // 2, w // 2
    for c, n, pool in ((16, 32, True), (32, 64, True), (64, 128, False)):
        tot += h * w * (9 * c + c * n)            # separable MACs (the reference's count)
        if pool:
            h, w = h // 2, w // 2
    return tot + 128 * 64 + 64 * 2


ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--kinds", default="residual,standard,small")
ap.add_argument("--dtypes", default="bf16_approx,fp32")
args = ap.parse_args()
bound_torch_threads()
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand((args.batch, 1, 90, 101), device=dev)
for kind in args.kinds.split(","):
    for dtype in args.dtypes.split(","):
        m = cda.create_model(kind, n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype).to(dev).eval()
        for _ in range(2):
            m(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            m(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        mm = macs(kind)
        print(f"{kind:9s} {dtype}: {ms:8.3f} ms / {args.batch} clips  {args.batch / ms / 1e3:8.3f} M clips/s  "
              f"{2 * mm * args.batch / ms / 1e9:8.1f} TFLOP/s (reference MAC count {mm / 1e6:.1f} M/clip)", flush=True)
        del m
