"""Diagnostic: phase breakdown (in-kernel s_memtime stamps) of the one-launch featuriser for several filterbanks.
Needs a library built with -DCOUGH_K1_STAMPS: bash tools/build_variant.sh stamps -DCOUGH_K1_STAMPS, then on the GPU box
K1_STAMPS_LIB=build_ab/lib_stamps.so python tools/k1_stamps_fullband.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cough_detector_amd import _lib, synth  # noqa: E402

_lib.LIB_PATH = os.path.abspath(os.environ.get("K1_STAMPS_LIB", os.path.join(ROOT, "build_ab", "lib_stamps.so")))
import cough_detector_amd as cda  # noqa: E402

NAMES = ["prologue", "P1 frames (wave 0)", "P1 wait for slowest wave", "P2 floor + mel rows out", "P2 DCT + mean + std",
         "P2 z-score + MFCC / delta rows out"]
SHIPPED = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)
lib = _lib.load()
lib.cough_debug_set_stamp_buffer.argtypes = [C.c_void_p]
B = 4096
wav = synth.device_clips(0, B)
for name, kw in (("shipped", {}), ("f_max 8000 / 64 mel", dict(f_max=8000.0)), ("80 mel / 20 MFCC", dict(n_mels=80, n_mfcc=20, f_max=8000.0)),
                 ("128 mel", dict(n_mels=128, f_max=8000.0))):
    pre = cda.AudioPreprocessor(device="cuda", **kw, **SHIPPED)
    stamps = torch.zeros(B * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        pre.featurize_batch(wav, normalize=True)
    assert lib.cough_debug_set_stamp_buffer(stamps.data_ptr()) == 0
    pre.featurize_batch(wav, normalize=True)
    torch.cuda.synchronize()
    assert lib.cough_debug_set_stamp_buffer(None) == 0
    st = stamps.view(B, 8).cpu().double()
    d = st[:, 1:7] - st[:, 0:6]
    total = st[:, 6] - st[:, 0]
    print(f"{name} [{pre.kernel_path()}]: median workgroup lifetime {total.median():.0f} cycles")
    for i, n in enumerate(NAMES):
        print(f"  {n:36s} median {d[:, i].median():9.0f}  share {100 * d[:, i].median() / total.median():5.1f}%")
