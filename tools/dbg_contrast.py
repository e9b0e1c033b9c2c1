import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import cough_detector_amd as cda
from oracle import featurizer as ofeat
from parity import synth_batch
w = synth_batch(0, 6, peak_normalize=False)
for nb in (1, 2, 3, 4, 6):
    flags = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=True, n_contrast_bands=nb)
    p = cda.AudioPreprocessor(device="cuda", **flags)
    for norm in (False, True):
        got = p.featurize_batch(w.cuda(), normalize=norm).cpu()[:, 90:]
        ref = ofeat.extract_features_batch(w, normalize_first=norm, **flags)[:, 90:]
        d = (got - ref).abs()
        print(nb, norm, 'per-row max err', [float('%.2e' % x) for x in d.amax(dim=(0, 2))], 'per-clip', [float('%.2e' % x) for x in d.amax(dim=(1, 2))])
