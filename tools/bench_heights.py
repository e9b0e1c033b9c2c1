"""Classifier time per 4096 clips from RESIDENT feature images of the heights the reference's flags produce (90 / 103 / 110
rows x 101 frames), exact-f32 vs split-bf16 kernels.  Run on the GPU box: python tools/bench_heights.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cough_detector_amd as cda
from cough_detector_amd import synth
from cough_detector_amd.hostcpu import bound_torch_threads

bound_torch_threads()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sd = synth.random_state_dict(seed=3)
print(f"lib {os.environ.get('COUGH_AMD_LIB', 'default')}, B = {B}")
for h in (64, 90, 95, 103, 110):
    xs = [torch.rand(B, 1, h, 101, device="cuda") for _ in range(3)]
    row = []
    for dtype in ("fp32", "bf16x3"):
        m = cda.create_model("residual", n_mels=h, compute_dtype=dtype)
        m.load_state_dict(sd)
        m.cuda().eval()
        for i in range(10):
            m(xs[i % 3])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            m(xs[i % 3])
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 30)
    print(f"{h}x101: fp32 {row[0]:.3f} ms  bf16x3 {row[1]:.3f} ms  ({row[0] / row[1]:.2f}x)  "
          f"effective {cda.create_model('residual', compute_dtype='bf16x3').effective_dtype(h, 101)}", flush=True)
