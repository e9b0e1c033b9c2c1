"""Featuriser and waveform -> logits throughput for the flag sets the reference itself produces, shipped geometry, 1 s clips:
the shipped set (src/train.py:264-287), delta-delta only (103 rows) and the constructor's / engine's missing-key defaults
(src/preprocessing.py:43-49, src/inference.py:100-106: PCEN, pre-emphasis, delta-delta, 6 contrast bands -> 110 rows).
Run on the GPU box: python tools/bench_flags.py [B]"""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cough_detector_amd as cda
from cough_detector_amd import synth
from cough_detector_amd.hostcpu import bound_torch_threads

bound_torch_threads()
warnings.simplefilter("ignore")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""      # substring filter on the case name (for a rocprofv3 run of one case)
OFF = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)
CASES = [("shipped flags (90 rows)", OFF),
         ("delta-delta (103 rows)", dict(OFF, use_delta_delta=True)),
         ("pre-emphasis (90 rows)", dict(OFF, use_pre_emphasis=True)),
         ("PCEN (90 rows)", dict(OFF, use_pcen=True)),
         ("pre-emphasis + PCEN + delta-delta (103 rows)", dict(OFF, use_delta_delta=True, use_pcen=True, use_pre_emphasis=True)),
         ("contrast, 4 bands (95 rows)", dict(OFF, use_spectral_contrast=True, n_contrast_bands=4)),
         ("constructor defaults: all on, 6 bands (110 rows)", dict())]
sd = synth.random_state_dict(seed=3)


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, kw in CASES:
    if ONLY not in name:
        continue
    pre = cda.AudioPreprocessor(device="cuda", **kw)
    rows = pre.get_num_features()
    w = torch.randn(B, pre.segment_samples, device="cuda") * 0.1
    out = torch.empty((B, rows, 101), device="cuda")
    f_ms = timed(lambda: pre.featurize_batch(w, normalize=True, out=out))
    model = cda.create_model("residual", n_mels=rows, compute_dtype="bf16x3")
    model.load_state_dict(sd)
    pipe = cda.CoughPipeline(pre, model.cuda().eval())
    p_ms = timed(lambda: pipe(w, normalize=True))
    print(f"{name:50s} featurise {f_ms:7.3f} ms ({B / f_ms / 1e3:6.2f} M clips/s)   waveform -> logits {p_ms:7.3f} ms "
          f"({B / p_ms / 1e3:6.2f} M clips/s, {model.effective_dtype(rows, 101)})", flush=True)
