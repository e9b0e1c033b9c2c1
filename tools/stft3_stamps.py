"""Diagnostic: phase breakdown of one steady-state iteration (the 6th clip of each workgroup) of the persistent STFT
kernel (stft3_kernel), from in-kernel s_memtime stamps of wave 0 and wave 3 (-DCOUGH_K1_STAMPS build; the product
build holds no stamp).  Run on the GPU box: K1_STAMPS_LIB=build_ab/libstamps.so python tools/stft3_stamps.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cough_detector_amd import _lib, synth  # noqa: E402

_lib.LIB_PATH = os.path.abspath(os.environ["K1_STAMPS_LIB"])
import cough_detector_amd as cda  # noqa: E402

NAMES = ["clip start -> round-0 samples landed (vmcnt)", "-> window, radix-16, twiddle, two transposes", "-> round-1 DMA issued",
         "-> radix-16 #2, split, powers in the image", "-> round 1 (the same four steps)", "-> barrier 1 (image complete)",
         "-> flush stores issued + barrier 2 (image free)"]
lib = _lib.load()
B = 4096
wav = synth.device_clips(0, B)
pre = cda.AudioPreprocessor(device="cuda", use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                            use_spectral_contrast=False)
spec = torch.empty((B, 257, 101), dtype=torch.float32, device="cuda")
for _ in range(20):
    pre.spectrogram_batch(wav, out=spec)
n_wg = 256
stamps = torch.zeros(n_wg * 2 * 8, dtype=torch.int64, device="cuda")
lib.cough_debug_set_stft_stamp_buffer.argtypes = [C.c_void_p]
assert lib.cough_debug_set_stft_stamp_buffer(stamps.data_ptr()) == 0
pre.spectrogram_batch(wav, out=spec)
torch.cuda.synchronize()
assert lib.cough_debug_set_stft_stamp_buffer(None) == 0
st = stamps.view(n_wg, 2, 8).cpu().double()
for w, wname in ((0, "wave 0"), (1, "wave 3")):
    rows = st[:, w]
    rows = rows[(rows > 0).all(dim=1)]
    d = rows[:, 1:8] - rows[:, 0:7]
    total = rows[:, 7] - rows[:, 0]
    print(f"{wname}: {rows.shape[0]} workgroups; iteration median {total.median():.0f} ticks (p10 {total.quantile(0.1):.0f}, p90 {total.quantile(0.9):.0f})")
    for i, n in enumerate(NAMES):
        print(f"    {n:52s} median {d[:, i].median():8.0f}  share {100 * d[:, i].median() / total.median():5.1f}%")
