"""Diagnostic: cost of the stand-alone split-bf16 stem kernel (features in HBM -> a1) next to the fused featurise + stem kernel.
Prints ms per 4096 clips of (a) classifier from materialised features (stem kernel + blocks), (b) featurise alone,
(c) the fused pipeline.  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from cough_detector_amd.hostcpu import bound_torch_threads


def timed(fn, n=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    bound_torch_threads()
    B = 4096
    pre = cda.AudioPreprocessor(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False,
                                device="cuda")
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(synth.random_state_dict(seed=3))
    model.cuda().eval()
    wav = synth.device_clips(0, B)
    feats = torch.empty((B, 90, 101), dtype=torch.float32, device="cuda")
    pre.featurize_batch(wav, normalize=True, out=feats)
    x = feats.unsqueeze(1)
    pipe = cda.CoughPipeline(pre, model)
    for _ in range(200):          # clocks
        pipe(wav)
    t_cls = timed(lambda: model(x))
    t_feat = timed(lambda: pre.featurize_batch(wav, normalize=True, out=feats))
    t_pipe = timed(lambda: pipe(wav))
    print(f"classifier from features (stem kernel + blocks) {t_cls:.4f} ms; featurise alone {t_feat:.4f} ms; "
          f"sum {t_cls + t_feat:.4f} ms; fused pipeline {t_pipe:.4f} ms")


if __name__ == "__main__":
    main()
