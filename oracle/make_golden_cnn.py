"""Generate ``tests/golden/cnn_golden.npz``: goldens of the reference's other two classifiers.

Run in the BUILD CONTAINER only (needs ``/root/reference``):  ``python -m oracle.make_golden_cnn``

Executes the REFERENCE modules ``CoughDetector`` ("standard") and ``CoughDetectorSmall`` ("small") of
``/root/reference/src/model.py`` (imported by file path; torch only) in eval mode on the first 8 feature images
of ``features_golden.npz``: seeded weights, BatchNorm statistics and affine parameters randomised (fresh 0/1 stats
would not exercise the folding), Linear weights x8 (default init leaves the class margins at ~1e-4), then the last
Linear scaled and re-centred to a TRAINED head's logit scale (class-margin std 2.5, as resnet_golden.npz), margin
median at 0 so that both classes occur.  Stored per model: the
state_dict, the conv-stack output before the global mean, logits, softmax, argmax.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.make_golden import OUT, load_reference_model_module   # noqa: E402

N = 8


def main():
    torch.set_num_threads(1)
    ref = load_reference_model_module()
    feats = np.load(os.path.join(OUT, "features_golden.npz"))["features"][:N]
    x = torch.from_numpy(feats).unsqueeze(1).contiguous()                     # (8, 1, 90, 101)
    out = {"x": x.numpy()}
    for kind, seed, last in (("standard", 11, "fc.3"), ("small", 13, "classifier.4")):
        torch.manual_seed(seed)
        net = ref.create_model(kind, n_mels=90, num_classes=2, in_channels=1).eval()
        g = torch.Generator().manual_seed(seed + 100)
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                c = m.num_features
                m.running_mean.copy_(torch.randn(c, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(c, generator=g) * 1.5 + 0.25)
                m.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
                m.bias.data.copy_(torch.randn(c, generator=g) * 0.2)
                m.num_batches_tracked.fill_(77)
            elif isinstance(m, torch.nn.Conv2d) and m.groups > 1:
                m.weight.data.mul_(2.0)                                        # depthwise taps: keep the signal alive
            elif isinstance(m, torch.nn.Linear):
                m.weight.data.mul_(8.0)                                        # default init gives margins ~1e-4
        with torch.no_grad():
            # trained-scale head (as oracle/make_golden.py): class-margin std 2.5, logits centred, margin median at 0
            lastm = dict(net.named_modules())[last]
            l = net(x)
            lastm.weight.data.mul_(2.5 / (l[:, 1] - l[:, 0]).std())
            lastm.bias.data.sub_(net(x).mean(dim=0))
            l = net(x)
            d = (l[:, 1] - l[:, 0]).sort().values
            lastm.bias.data[1] -= 0.5 * (d[N // 2 - 1] + d[N // 2])
            conv_out = net.conv_layers(x) if kind == "standard" else net.features[:-1](x)
            logits = net(x)
            preds, probs = net.predict(x)
        out.update({f"{kind}.conv_out": conv_out.numpy(), f"{kind}.logits": logits.numpy(),
                    f"{kind}.probs": probs.numpy(), f"{kind}.preds": preds.numpy()})
        out.update({f"{kind}.sd.{k}": v.detach().numpy() for k, v in net.state_dict().items()})
        print(kind, "params", sum(p.numel() for p in net.parameters()), "conv_out", tuple(conv_out.shape))
        print(" logits", logits.numpy().round(4).tolist(), "preds", preds.tolist())
    path = os.path.join(OUT, "cnn_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
