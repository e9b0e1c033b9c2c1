"""Diagnostic: where a workgroup of the STFT kernel (cough_spectrogram) spends its life, from in-kernel s_memtime
stamps of wave 0 and wave 3.  Builds a SEPARATE library (-DCOUGH_K1_STAMPS); the product build holds no stamp.
Run on the GPU box:  python tools/stft_stamps.py
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
NAMES = ["entry -> tables in LDS + barrier", "-> first group's samples arrived (window done)", "-> first group transformed, powers in LDS",
         "-> second group done (loop exit)", "-> workgroup barrier (wait for slowest wave)", "-> flush stores issued",
         "-> flush stores complete (vmcnt 0)"]


def main():
    global LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if os.environ.get("K1_STAMPS_LIB"):
        LIB = os.path.abspath(os.environ["K1_STAMPS_LIB"])
    else:
        cmd = [build._hipcc(), *build.FLAGS, "-DCOUGH_K1_STAMPS", "-o", LIB] + \
              [os.path.join(build.CSRC, s) for s in build.SOURCES]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    _lib.LIB_PATH = LIB
    import cough_detector_amd as cda
    lib = _lib.load()
    B = int(os.environ.get("STFT_STAMPS_B", "4096"))
    wav = synth.device_clips(0, B)
    pre = cda.AudioPreprocessor(device="cuda", use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                                use_spectral_contrast=False)
    spec = torch.empty((B, 257, 101), dtype=torch.float32, device="cuda")
    for _ in range(20):
        pre.spectrogram_batch(wav, out=spec)
    n_wg = ((B + 7) // 8) * 8 * 4
    stamps = torch.zeros(n_wg * 2 * 8, dtype=torch.int64, device="cuda")
    lib.cough_debug_set_stft_stamp_buffer.argtypes = [C.c_void_p]
    assert lib.cough_debug_set_stft_stamp_buffer(stamps.data_ptr()) == 0
    pre.spectrogram_batch(wav, out=spec)
    torch.cuda.synchronize()
    assert lib.cough_debug_set_stft_stamp_buffer(None) == 0
    st = stamps.view(n_wg, 2, 8).cpu().double()
    chunk = (torch.arange(n_wg) // 8) % 4
    span = st[:, :, 7].max() - st[:, :, 0].min()
    print(f"kernel span {span:.0f} ticks over {n_wg} workgroups; {n_wg / 256:.1f} workgroups per CU")
    for w, wname in ((0, "wave 0"), (1, "wave 3")):
        for ch in range(4):
            rows = st[chunk == ch, w]
            d = rows[:, 1:8] - rows[:, 0:7]
            total = rows[:, 7] - rows[:, 0]
            print(f"{wname}, chunk {ch} ({[7, 7, 6, 6][ch]} groups): median lifetime {total.median():.0f} ticks "
                  f"(p10 {total.quantile(0.1):.0f}, p90 {total.quantile(0.9):.0f})")
            for i, n in enumerate(NAMES):
                print(f"    {n:52s} median {d[:, i].median():8.0f}  share {100 * d[:, i].median() / total.median():5.1f}%")
    life = (st[:, :, 7].max(dim=1).values - st[:, :, 0].min(dim=1).values)
    print(f"workgroup lifetime median {life.median():.0f} ticks; sum of lifetimes / (span x 256 CUs) = "
          f"{life.sum() / (span * 256):.2f} workgroups resident per CU on average")


if __name__ == "__main__":
    main()
