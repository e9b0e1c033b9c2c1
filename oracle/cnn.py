"""Functional CPU restatement of the two other classifiers (TEST INFRASTRUCTURE, not product).

Follows ``/root/reference/src/model.py``: ``ConvBlock.forward`` (:34-40), ``CoughDetector.forward`` /
``predict`` (:107-141) and ``CoughDetectorSmall`` (:144-207), eval mode (BatchNorm on running stats, eps 1e-5,
Dropout / Dropout2d = identity).  Takes the reference's own ``state_dict``.
PINNED: ``tests/test_oracle_cnn.py`` checks it against goldens produced by running the reference modules
themselves (``oracle/make_golden_cnn.py``).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

EPS = 1e-5


def _bn(x, sd, prefix):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training=False, eps=EPS)


def standard_features(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """conv_layers: ConvBlock x len(channels): conv3x3 p1 -> BN -> ReLU -> MaxPool2d(2) (model.py:24-40, :86-92)."""
    i = 0
    while f"conv_layers.{i}.conv.weight" in sd:
        p = f"conv_layers.{i}"
        x = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        x = F.max_pool2d(F.relu(_bn(x, sd, p + ".bn")), 2)
        i += 1
    return x


def standard_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """CoughDetector.forward (model.py:107-127): conv blocks -> global mean -> Linear -> ReLU -> Linear."""
    v = standard_features(x, sd).mean(dim=(2, 3))
    h = F.relu(F.linear(v, sd["fc.0.weight"], sd["fc.0.bias"]))
    return F.linear(h, sd["fc.3.weight"], sd["fc.3.bias"])


def small_features(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """CoughDetectorSmall.features (model.py:162-188): conv3x3 -> BN -> ReLU -> pool, then three depthwise-separable
    blocks (depthwise 3x3 groups=C, pointwise 1x1, BN, ReLU; MaxPool2d(2) after the first two, global mean after
    the third).  Returned BEFORE the global mean: (B, 128, h, w)."""
    f = "features."
    x = F.conv2d(x, sd[f + "0.weight"], sd[f + "0.bias"], padding=1)
    x = F.max_pool2d(F.relu(_bn(x, sd, f + "1")), 2)
    for dw, pw, bn, pool in ((4, 5, 6, True), (9, 10, 11, True), (14, 15, 16, False)):
        c = sd[f"{f}{dw}.weight"].shape[0]
        x = F.conv2d(x, sd[f"{f}{dw}.weight"], sd[f"{f}{dw}.bias"], padding=1, groups=c)
        x = F.conv2d(x, sd[f"{f}{pw}.weight"], sd[f"{f}{pw}.bias"])
        x = F.relu(_bn(x, sd, f"{f}{bn}"))
        if pool:
            x = F.max_pool2d(x, 2)
    return x


def small_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """CoughDetectorSmall.forward (model.py:198-201): features -> Flatten -> Linear(128,64) -> ReLU -> Linear(64,2)."""
    v = small_features(x, sd).mean(dim=(2, 3))
    h = F.relu(F.linear(v, sd["classifier.1.weight"], sd["classifier.1.bias"]))
    return F.linear(h, sd["classifier.4.weight"], sd["classifier.4.bias"])


FORWARD = {"standard": standard_forward, "small": small_forward}
FEATURES = {"standard": standard_features, "small": small_features}


def predict(kind: str, x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """predict (model.py:129-141, :203-207)."""
    probs = F.softmax(FORWARD[kind](x, sd), dim=1)
    return probs.argmax(dim=1), probs
