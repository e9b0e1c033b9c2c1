"""Experiment: can the VALU-bound featurise kernel and the MFMA-bound residual-block kernels share
CUs?  Three K1 workgroups (45.5 KB of LDS each) fill a CU, so on two plain streams the block workgroups (74-80 KB) only
enter when K1 drains.  This build (-DCOUGH_EXP_OVERLAP) lets K1 CLAIM more LDS per workgroup: at 80 KB a CU holds two
slots, each a K1 or a block workgroup, and a block workgroup of batch i can sit beside a K1 workgroup of batch i + 1.
Stream A runs K1 (+ stem) of the next batch while stream B runs blocks + head of the current one.
Run on the GPU box:  COUGH_AMD_LIB=build_ab/libexp.so python tools/exp_coresident.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cough_detector_amd import AudioPreprocessor, CoughPipeline, create_model, synth
from cough_detector_amd.hostcpu import bound_torch_threads

bound_torch_threads()
dev = torch.device("cuda:0")
B, STEPS = 4096, 60
pre = AudioPreprocessor(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False,
                        device="cuda")
model = create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16x3")
model.load_state_dict(synth.random_state_dict(seed=3))
model.to(dev).eval()
pool = torch.empty((3 * B, 16000), dtype=torch.float32, device=dev)
for r in range(3):
    synth.device_clips(r * B, B, out=pool[r * B:(r + 1) * B])
batches = [pool[r * B:(r + 1) * B] for r in range(3)]
pipes = [CoughPipeline(pre, model) for _ in range(2)]         # two workspaces: a1 of batch i and of batch i + 1
for p in pipes:
    p(batches[0])
torch.cuda.synchronize()


def timed(fn, n=STEPS):
    for _ in range(10):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def setenv(**kw):
    for k in ("COUGH_EXP_ONLY", "COUGH_EXP_K1_LDS"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


setenv()
print(f"serial pipeline, one stream:                        {timed(lambda i: pipes[0](batches[i % 3])):.4f} ms/step", flush=True)
for lds in (0, 54 * 1024, 80 * 1024, 81 * 1024):
    setenv(COUGH_EXP_ONLY="k", **({"COUGH_EXP_K1_LDS": lds} if lds else {}))
    print(f"K1 + stem alone, LDS claim {lds // 1024:3d} KB ({'3' if lds == 0 else 160 // max(lds // 1024, 1)} workgroups per CU): "
          f"{timed(lambda i: pipes[0](batches[i % 3])):.4f} ms", flush=True)
setenv(COUGH_EXP_ONLY="b")
print(f"blocks + head alone:                                 {timed(lambda i: pipes[0](batches[i % 3])):.4f} ms", flush=True)

sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for lds in (0, 54 * 1024, 80 * 1024):
    def step(i):
        # batch i + 1 is featurised on stream A into workspace (i + 1) % 2 while batch i is classified on stream B
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(sa):
            setenv(COUGH_EXP_ONLY="k", **({"COUGH_EXP_K1_LDS": lds} if lds else {}))
            pipes[(i + 1) % 2](batches[(i + 1) % 3])
            ea.record(sa)
        with torch.cuda.stream(sb):
            setenv(COUGH_EXP_ONLY="b")
            pipes[i % 2](batches[i % 3])
            eb.record(sb)
        sa.wait_event(eb)       # next step's K1 overwrites the workspace the blocks just read
        sb.wait_event(ea)       # next step's blocks read what this K1 wrote
    ms = timed(step)
    print(f"two streams, K1(i+1) || blocks(i), K1 LDS claim {lds // 1024:3d} KB: {ms:.4f} ms/step  ({B / ms / 1e3:.3f} M clips/s)", flush=True)
setenv()
