"""Generate ``tests/golden/resnet_channels_golden.npz`` (reference outputs for NON-default ``channels`` tuples) and
``tests/golden/resblock_golden.npz`` (the reference ``ResidualBlock`` called on its own, identity skip included).

Run in the BUILD CONTAINER only (needs ``/root/reference``):

    python -m oracle.make_golden_channels

``CoughDetectorResidual(channels=...)`` (/root/reference/src/model.py:216-247) builds one stride-2 projection
block per consecutive pair of the tuple.  The shipped checkpoint uses (32, 64, 128); this fixture pins the oracle
and the HIP path on tuples the fused kernels do not cover: a deeper, non-multiple-of-32 one and a single block with
in_channels == out_channels (stride 2 still makes the skip a 1x1 projection, model.py:280-283).  Inputs are the first
clips of ``features_golden.npz`` (not stored again); the head is calibrated to a trained detector's logit scale
exactly as ``oracle/make_golden.py`` does.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.make_golden import MARGIN_STD, OUT, load_reference_model_module      # noqa: E402

CASES = {"deep": (16, 24, 40, 72), "square": (48, 48)}
N_CLIPS = 6
# ResidualBlock used directly (model.py:268-293): (in_channels, out_channels, stride); "identity" is the nn.Identity skip
BLOCKS = {"identity": (16, 16, 1), "proj_s1": (24, 40, 1), "proj_s2": (32, 64, 2), "same_s2": (48, 48, 2)}
BLOCK_INPUT = (3, 13, 17)      # batch, height, width


def main():
    torch.set_num_threads(1)
    ref = load_reference_model_module()
    feats = np.load(os.path.join(OUT, "features_golden.npz"))["features"][:N_CLIPS]
    x = torch.from_numpy(feats).unsqueeze(1).contiguous()
    out = {}
    for name, channels in CASES.items():
        torch.manual_seed(sum(channels) * 7919)
        net = ref.create_model("residual", n_mels=90, num_classes=2, in_channels=1, channels=channels).eval()
        g = torch.Generator().manual_seed(11 + len(channels))
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                c = m.num_features
                m.running_mean.copy_(torch.randn(c, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(c, generator=g) * 1.5 + 0.25)
                m.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
                m.bias.data.copy_(torch.randn(c, generator=g) * 0.2)
        with torch.no_grad():
            l = net(x)
            net.fc[2].weight.data.mul_(MARGIN_STD / (l[:, 1] - l[:, 0]).std())
            net.fc[2].bias.data.sub_(net(x).mean(dim=0))
            l = net(x)
            d = (l[:, 1] - l[:, 0]).sort().values
            net.fc[2].bias.data[1] -= 0.5 * (d[N_CLIPS // 2 - 1] + d[N_CLIPS // 2])
            a = net.conv1(x)
            out[f"{name}.a1"] = a.numpy()
            for blk in net.res_blocks:
                a = blk(a)
            out[f"{name}.a_last"] = a.numpy()
            logits = net(x)
            preds, probs = net.predict(x)
        out[f"{name}.channels"] = np.array(channels, dtype=np.int32)
        out[f"{name}.logits"] = logits.numpy()
        out[f"{name}.probs"] = probs.numpy()
        out[f"{name}.preds"] = preds.numpy()
        for k, v in net.state_dict().items():
            out[f"{name}.sd.{k}"] = v.detach().numpy()
        print(name, channels, "logits", logits.numpy().round(3).tolist(), "preds", preds.tolist())
    path = os.path.join(OUT, "resnet_channels_golden.npz")
    np.savez_compressed(path, n_clips=N_CLIPS, **out)
    print(path, os.path.getsize(path))

    blocks = {}
    for name, (cin, cout, stride) in BLOCKS.items():
        torch.manual_seed(1000 + cin * 7 + cout * 3 + stride)
        blk = ref.ResidualBlock(cin, cout, stride=stride).eval()
        g = torch.Generator().manual_seed(cin + cout + stride)
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                c = m.num_features
                m.running_mean.copy_(torch.randn(c, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(c, generator=g) * 1.5 + 0.25)
                m.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
                m.bias.data.copy_(torch.randn(c, generator=g) * 0.2)
        xb = torch.randn(BLOCK_INPUT[0], cin, BLOCK_INPUT[1], BLOCK_INPUT[2], generator=g)
        with torch.no_grad():
            yb = blk(xb)
        blocks[f"{name}.cfg"] = np.array([cin, cout, stride], dtype=np.int32)
        blocks[f"{name}.x"] = xb.numpy()
        blocks[f"{name}.y"] = yb.numpy()
        for k, v in blk.state_dict().items():
            blocks[f"{name}.sd.{k}"] = v.detach().numpy()
        print("block", name, (cin, cout, stride), "identity skip" if isinstance(blk.skip, torch.nn.Identity) else "projection",
              tuple(yb.shape), float(yb.abs().max()))
    path = os.path.join(OUT, "resblock_golden.npz")
    np.savez_compressed(path, **blocks)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
