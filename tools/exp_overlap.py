"""Experiment: does featurise(chunk i+1) overlap classifier(chunk i) when the batch is split over HIP streams?

Usage: python tools/exp_overlap.py [--batch 4096] [--steps 40]
Prints ms/step for 1 stream x 1 chunk (the shipped path) and for S streams x C chunks per stream.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cough_detector_amd import CoughPipeline, AudioPreprocessor, create_model
from cough_detector_amd.hostcpu import bound_torch_threads
from cough_detector_amd.synth import make_clips, random_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    bound_torch_threads()
    dev = torch.device("cuda:0")
    pre = AudioPreprocessor(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                            use_spectral_contrast=False, device="cuda")
    model = create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=os.environ.get("COUGH_EXP_DTYPE", "bf16x3"))
    model.load_state_dict(random_state_dict(seed=3))
    model.to(dev).eval()
    wav = torch.from_numpy(make_clips(0, args.batch, peak_normalize=False)).to(dev)
    ref = CoughPipeline(pre, model)(wav).clone()

    def run(n_streams, n_chunks, prio=False):
        parts = n_streams * n_chunks
        # uneven splits (e.g. 4096 over 3 parts) hand the remainder to the first parts: every clip is scored
        cuts = [args.batch * p // parts for p in range(parts + 1)]
        # prio: stream 0 outranks stream 1 (outranks 2 ...): the dispatcher fills a kernel's tail with the next stream's
        # workgroups instead of running the halves side by side
        lo, hi = torch.cuda.Stream.priority_range()
        pr = [max(hi, min(lo, hi + k)) for k in range(n_streams)] if prio else [0] * n_streams
        streams = [torch.cuda.Stream(dev, priority=pr[k]) for k in range(n_streams)]
        pipes = [CoughPipeline(pre, model) for _ in range(parts)]
        outs = [None] * parts

        def step():
            main_s = torch.cuda.current_stream(dev)
            for s in streams:
                s.wait_stream(main_s)
            for p in range(parts):
                s = streams[p % n_streams]
                with torch.cuda.stream(s):
                    outs[p] = pipes[p](wav[cuts[p]:cuts[p + 1]])
            for s in streams:
                main_s.wait_stream(s)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        got = torch.cat(outs)
        same = bool((got == ref).all())
        print(f"streams={n_streams} chunks/stream={n_chunks} prio={pr}  {ms:.4f} ms/step  {args.batch / ms / 1e3:.3f} M clips/s  "
              f"bit-identical={same}", flush=True)

    print("priority range (lowest, highest):", torch.cuda.Stream.priority_range())
    for cfg in [(1, 1), (2, 1), (2, 1, True), (2, 2, True), (4, 1, True), (2, 1), (2, 1, True), (1, 1)]:
        run(*cfg)


if __name__ == "__main__":
    main()
