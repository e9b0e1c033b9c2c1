"""Functional CPU restatement of the classifier (TEST INFRASTRUCTURE, not product).

Follows ``/root/reference/src/model.py``: ``CoughDetectorResidual.forward``
(:253-259), ``ResidualBlock.forward`` (:285-293), ``predict`` (:261-265), with
eval-mode BatchNorm (running stats, eps 1e-5) and Dropout = identity.  Takes
the reference's own ``state_dict`` (keys listed in SURVEY.md section 8a M0).
PINNED: ``tests/test_oracle_resnet.py`` checks it against goldens produced by
running the reference module itself (``oracle/make_golden.py``).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

EPS = 1e-5


def _bn(x, sd, prefix):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training=False, eps=EPS)


def stem(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """M1 -- model.py:227-232: conv7x7 s2 p3 -> BN -> ReLU -> maxpool 2."""
    y = F.conv2d(x, sd["conv1.0.weight"], sd["conv1.0.bias"], stride=2, padding=3)
    return F.max_pool2d(F.relu(_bn(y, sd, "conv1.1")), 2)


def res_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], i: int) -> torch.Tensor:
    """M2/M3 -- model.py:285-293 with the projection skip of :280-283."""
    p = f"res_blocks.{i}"
    identity = _bn(F.conv2d(x, sd[p + ".skip.0.weight"], sd[p + ".skip.0.bias"], stride=2), sd, p + ".skip.1")
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride=2, padding=1), sd, p + ".bn1"))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1), sd, p + ".bn2")
    return F.relu(out + identity)


def res_block_module(x: torch.Tensor, sd: Dict[str, torch.Tensor], stride: int) -> torch.Tensor:
    """One ResidualBlock on its own (model.py:268-293), state_dict keys of the block itself: the skip is the 1x1
    stride-s conv + BN when the block has one (``skip.0.weight`` present) and the identity otherwise (:280-283)."""
    if "skip.0.weight" in sd:
        identity = _bn(F.conv2d(x, sd["skip.0.weight"], sd["skip.0.bias"], stride=stride), sd, "skip.1")
    else:
        identity = x
    out = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], stride=stride, padding=1), sd, "bn1"))
    out = _bn(F.conv2d(out, sd["conv2.weight"], sd["conv2.bias"], padding=1), sd, "bn2")
    return F.relu(out + identity)


def head(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """M4 -- model.py:242-247,257-258: global mean -> Linear(128, 2)."""
    return F.linear(x.mean(dim=(2, 3)), sd["fc.2.weight"], sd["fc.2.bias"])


def forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], return_intermediates: bool = False):
    """Any ``channels`` tuple (model.py:216-247): one block per ``res_blocks.i`` found in the state_dict."""
    acts = [stem(x, sd)]
    i = 0
    while f"res_blocks.{i}.conv1.weight" in sd:
        acts.append(res_block(acts[-1], sd, i))
        i += 1
    logits = head(acts[-1], sd)
    if return_intermediates:
        return logits, tuple(acts)
    return logits


def predict(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """M5 -- model.py:261-265."""
    probs = F.softmax(forward(x, sd), dim=1)
    return probs.argmax(dim=1), probs
