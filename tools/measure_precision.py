#!/usr/bin/env python3
"""Logit error of every compute dtype of the HIP classifier on the reference-generated goldens
(tests/golden/resnet_golden.npz: trained-scale head, margin std 2.5).  GPU box; prints one line per dtype."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cough_detector_amd as cda  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "resnet_golden.npz"))
sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
x, ref = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["logits"])
margin = (ref[:, 1] - ref[:, 0])
print(f"goldens: {len(ref)} clips, margin std {margin.std():.3f}, min |margin| {margin.abs().min():.4f}, "
      f"max |logit| {ref.abs().max():.3f}")
for dtype in sys.argv[1:] or ["fp32", "bf16x3", "bf16_approx"]:
    m = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype)
    m.load_state_dict(sd)
    m.cuda().eval()
    logits = m(x).cpu()
    err = (logits - ref).abs().max().item()
    same = int((logits.argmax(1) == ref.argmax(1)).sum())
    acts = []
    for which, key in ((1, "a1"), (2, "a2"), (3, "a3")):
        a = m.read_activation(which).cpu()
        acts.append(f"{key} {float((a - torch.from_numpy(g[key])).abs().max()):.2e}")
    print(f"{dtype}: logits max abs err {err:.3e}, argmax equal {same}/{len(ref)}, " + ", ".join(acts))
