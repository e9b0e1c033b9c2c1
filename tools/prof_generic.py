"""One case of the generic featuriser chain, for rocprofv3 --kernel-trace --stats:
   rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_generic.py defaults2s"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings

import cough_detector_amd as cda

warnings.simplefilter("ignore")

OFF = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)
CASES = {"defaults2s": (dict(segment_duration=2.0, use_pcen=True, use_pre_emphasis=True, use_delta_delta=True,
                             use_spectral_contrast=True, n_contrast_bands=4), 2048),
         "defaults1s": (dict(), 4096),
         "contrast4": (dict(OFF, use_spectral_contrast=True, n_contrast_bands=4), 4096),
         "pcen1s": (dict(OFF, use_delta_delta=True, use_pcen=True, use_pre_emphasis=True), 4096),
         "fmax8k": (dict(f_max=8000.0, **OFF), 4096),
         "nfft1024": (dict(n_fft=1024, **OFF), 4096),
         "nfft400": (dict(n_fft=400, **OFF), 4096)}
kw, b = CASES[sys.argv[1]]
pre = cda.AudioPreprocessor(device="cuda", **kw)
w = torch.randn(b, pre.segment_samples, device="cuda") * 0.1
out = torch.empty((b, pre.get_num_features(), pre._frames(pre.segment_samples)), device="cuda")
for _ in range(12):
    pre.featurize_batch(w, normalize=True, out=out)
torch.cuda.synchronize()
