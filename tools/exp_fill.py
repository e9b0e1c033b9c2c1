"""HBM write / copy rates of plain torch kernels, for calibration.  Usage: python tools/exp_fill.py"""
import torch
dev = torch.device("cuda:0")
for mb in (425, 850):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn, nbytes in (("fill", lambda: a.fill_(1.0), 4 * n), ("copy", lambda: b.copy_(a), 8 * n)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name} {mb} MB: {ms:.4f} ms  {nbytes / ms / 1e6:.0f} GB/s")
