"""CPU restatement of SpecAugment (TEST INFRASTRUCTURE, not product).

Follows ``/root/reference/src/augmentation.py:271-331`` (``SpecAugment.__call__``) and the torchaudio transforms
it calls, ``T.FrequencyMasking(freq_mask_param)`` / ``T.TimeMasking(time_mask_param)`` -- torchaudio
``functional.mask_along_axis`` with ``iid_masks=False, p=1.0, mask_value=0`` restated from its published algorithm:

    value = torch.rand(1) * mask_param;  min_value = torch.rand(1) * (size - value)
    mask_start = min_value.long();  mask_end = min_value.long() + value.long()
    positions start <= i < end along the axis are filled with 0 for every item of the batch

PARITY UNPINNED against torchaudio (not installed, no reference test vectors); the draw order (python ``random``
for the coin, then two ``torch.rand(1)`` per mask on the CPU generator) is the reference's.
"""
from __future__ import annotations

import random
from typing import List, Tuple

import torch


def draw_mask(mask_param: int, size: int) -> Tuple[int, int]:
    """One mask_along_axis draw: (start, end) with end - start < mask_param."""
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (size - value)
    start = int(min_value.long())
    end = int(min_value.long() + value.long())
    return start, end


def mask_along_axis(spec: torch.Tensor, mask_param: int, axis: int) -> torch.Tensor:
    """axis counted from the end: -2 = frequency, -1 = time.  Returns a new tensor."""
    if mask_param < 1:
        return spec
    start, end = draw_mask(mask_param, spec.shape[axis])
    idx = torch.arange(spec.shape[axis])
    m = (idx >= start) & (idx < end)
    if axis == -2:
        m = m.unsqueeze(-1)
    return spec.masked_fill(m, 0.0)


def spec_augment(spec: torch.Tensor, freq_mask_param: int = 10, time_mask_param: int = 20, n_freq_masks: int = 2,
                 n_time_masks: int = 2, p: float = 0.5) -> torch.Tensor:
    """SpecAugment.__call__ (augmentation.py:303-331): (C, F, T) or (B, C, F, T)."""
    if random.random() > p:
        return spec
    for _ in range(n_freq_masks):
        spec = mask_along_axis(spec, freq_mask_param, -2)
    for _ in range(n_time_masks):
        spec = mask_along_axis(spec, time_mask_param, -1)
    return spec
