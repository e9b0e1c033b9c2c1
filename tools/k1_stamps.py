"""Diagnostic: phase breakdown of the featurise kernel from in-kernel s_memtime stamps.

Builds a SEPARATE library (libcough_amd_stamps.so, -DCOUGH_K1_STAMPS) so the product build never
contains a stamp.  Reports the share of each phase, not absolute run time (stamped builds are slower).
Run on the GPU box:  python tools/k1_stamps.py
"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cough_detector_amd import _lib, build, synth  # noqa: E402

LIB = os.path.join(ROOT, "gpurun_out", "libcough_amd_stamps.so")
NAMES = ["P0 peak (normalize)", "P1 frames (wave 0)", "P1 wait for slowest wave", "P2 floor+mel rows out",
         "P2 DCT+mean", "P2 std+zscore+delta out"]


def main():
    global LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if os.environ.get("K1_STAMPS_LIB"):       # prebuilt with -DCOUGH_K1_STAMPS (tools/build_variant.sh)
        LIB = os.path.abspath(os.environ["K1_STAMPS_LIB"])
    else:
        cmd = [build._hipcc(), *build.FLAGS, "-DCOUGH_K1_STAMPS", "-o", LIB] + \
              [os.path.join(build.CSRC, s) for s in build.SOURCES]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    _lib.LIB_PATH = LIB
    import cough_detector_amd as cda
    lib = _lib.load()
    B = 4096
    wav = torch.from_numpy(synth.make_clips(0, B, peak_normalize=False)).cuda()
    pre = cda.AudioPreprocessor(device="cuda", use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                                use_spectral_contrast=False)
    for normalize in (False, True):
        stamps = torch.zeros(B * 8, dtype=torch.int64, device="cuda")
        pre.featurize_batch(wav, normalize=normalize)
        lib.cough_debug_set_stamp_buffer.argtypes = [C.c_void_p]
        assert lib.cough_debug_set_stamp_buffer(stamps.data_ptr()) == 0
        pre.featurize_batch(wav, normalize=normalize)
        torch.cuda.synchronize()
        assert lib.cough_debug_set_stamp_buffer(None) == 0
        st = stamps.view(B, 8).cpu().double()
        d = st[:, 1:7] - st[:, 0:6]
        total = (st[:, 6] - st[:, 0])
        print(f"normalize={normalize}: per-workgroup cycles (s_memtime ticks @100MHz? see below) "
              f"median total {total.median():.0f}")
        for i, n in enumerate(NAMES):
            print(f"  {n:32s} median {d[:, i].median():9.0f}  share {100 * d[:, i].median() / total.median():5.1f}%")
        span = st[:, 6].max() - st[:, 0].min()
        print(f"  grid span {span:.0f} ticks; blocks {B}; mean per-block total {total.mean():.0f}")
    # fused pipeline (featurise + split-bf16 stem in one kernel): slot 7 = end of the stem phase
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(synth.random_state_dict(seed=3))
    model.cuda().eval()
    pipe = cda.CoughPipeline(pre, model)
    stamps = torch.zeros(B * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        pipe(wav)
    assert lib.cough_debug_set_stamp_buffer(stamps.data_ptr()) == 0
    pipe(wav)
    torch.cuda.synchronize()
    assert lib.cough_debug_set_stamp_buffer(None) == 0
    st = stamps.view(B, 8).cpu().double()
    d = st[:, 1:8] - st[:, 0:7]
    total = st[:, 7] - st[:, 0]
    print(f"fused featurise + stem (bf16x3): median workgroup lifetime {total.median():.0f}")
    for i, n in enumerate(NAMES + ["stem: hi/lo image already built; MFMA + pool + store"]):
        print(f"  {n:52s} median {d[:, i].median():9.0f}  share {100 * d[:, i].median() / total.median():5.1f}%")


if __name__ == "__main__":
    main()
