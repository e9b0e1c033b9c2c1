"""CPU emulation of the classifier's reduced-precision operand schemes (TEST INFRASTRUCTURE).

The HIP classifier multiplies on the matrix cores with f32 accumulation; what differs between its
compute dtypes is how the two operands of every product (BN-folded conv weights, activations) are
represented:

``bf16``     one bf16 value per operand (8 significant bits).
``bf16x3``   split-bf16: x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significant bits);
             a product is hi*hi + hi*lo + lo*hi -- three MFMAs into one f32 accumulator, the
             lo*lo term (2^-18 relative) is dropped.

This module restates that arithmetic with torch float64 convolutions over operands rounded the same
way (f32 accumulation error is ~1e-7 relative and ignored), so the error budget of a scheme can be
checked against the reference-generated goldens without a GPU (``tests/test_precision_budget.py``).
BatchNorm folding follows ``csrc/resnet.hip: fold`` (= model.py eval-mode BN, eps 1e-5).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

EPS = 1e-5


def bf16(x: torch.Tensor) -> torch.Tensor:
    return x.float().bfloat16().double()


def split(x: torch.Tensor):
    hi = bf16(x)
    return hi, bf16(x - hi)


def _fold(sd, conv, bn):
    s = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + EPS)
    w = sd[conv + ".weight"].double() * s[:, None, None, None]
    b = (sd[conv + ".bias"].double() - sd[bn + ".running_mean"].double()) * s + sd[bn + ".bias"].double()
    return w.float().double(), b.float().double()      # the library stores folded weights / biases as f32


def _conv(x, w, scheme, **kw):
    if scheme == "f32":
        return F.conv2d(x, w, **kw)
    if scheme == "bf16":
        return F.conv2d(bf16(x), bf16(w), **kw)
    xh, xl = split(x)
    wh, wl = split(w)
    return F.conv2d(xh, wh, **kw) + F.conv2d(xh, wl, **kw) + F.conv2d(xl, wh, **kw)


def _store(x, scheme):
    """Activation as it is kept between layers (LDS / HBM)."""
    if scheme == "f32":
        return x
    if scheme == "bf16":
        return bf16(x)
    hi, lo = split(x)
    return hi + lo


def forward(x: torch.Tensor, sd, scheme: str) -> torch.Tensor:
    """(B,1,H,W) features -> (B,2) logits under operand scheme ``f32`` | ``bf16`` | ``bf16x3``."""
    x = x.double()
    w, b = _fold(sd, "conv1.0", "conv1.1")
    a = F.max_pool2d(F.relu(_conv(x, w, scheme, stride=2, padding=3) + b[None, :, None, None]), 2)
    a = _store(a, scheme)
    for i in range(2):
        p = f"res_blocks.{i}"
        w1, b1 = _fold(sd, p + ".conv1", p + ".bn1")
        w2, b2 = _fold(sd, p + ".conv2", p + ".bn2")
        ws, bs = _fold(sd, p + ".skip.0", p + ".skip.1")
        h = _store(F.relu(_conv(a, w1, scheme, stride=2, padding=1) + b1[None, :, None, None]), scheme)
        y = _conv(h, w2, scheme, padding=1) + _conv(a, ws, scheme, stride=2) + (b2 + bs)[None, :, None, None]
        a = _store(F.relu(y), scheme)
    return (F.linear(a.mean(dim=(2, 3)), sd["fc.2.weight"].double(), sd["fc.2.bias"].double())).float()
