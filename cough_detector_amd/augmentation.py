"""SpecAugment on the MI355X, with the reference's interface (``/root/reference/src/augmentation.py:271-331``).

Same constructor arguments and call convention as the reference class; the random draws happen on the host in the
reference's order -- ``random.random()`` for the coin, then per mask two ``torch.rand(1)`` as torchaudio's
``mask_along_axis`` draws them -- so a seeded run masks the same rows / columns as the reference.  The masking itself
is one pass of ``cough_mask_axes`` over the whole batch (all frequency and time masks at once) instead of one
``masked_fill`` pass per mask.  Waveform-domain augmentation (``AudioAugmentor``, ``MixUp``) belongs to the training
loop and is not part of this build.
"""
from __future__ import annotations

import ctypes as C
import random
from typing import List, Tuple

import torch

from . import _lib
from .preprocessing import _cuda_device


class SpecAugment:
    def __init__(self, freq_mask_param: int = 10, time_mask_param: int = 20, n_freq_masks: int = 2,
                 n_time_masks: int = 2, p: float = 0.5):
        self.freq_mask_param, self.time_mask_param = freq_mask_param, time_mask_param
        self.n_freq_masks, self.n_time_masks = n_freq_masks, n_time_masks
        self.p = p
        if n_freq_masks + n_time_masks > 16:
            raise ValueError("SpecAugment: at most 16 masks per call on the MI355X path")

    @staticmethod
    def _draw(mask_param: int, size: int) -> Tuple[int, int]:
        # torchaudio.functional.mask_along_axis (iid_masks=False, p=1.0)
        value = torch.rand(1) * mask_param
        min_value = torch.rand(1) * (size - value)
        return int(min_value.long()), int(min_value.long() + value.long())

    def draw_masks(self, n_freq: int, n_time: int) -> List[Tuple[int, int, int]]:
        """[(axis, start, end)] in the reference's draw order: frequency masks, then time masks."""
        masks = []
        if self.freq_mask_param >= 1:
            masks += [(0,) + self._draw(self.freq_mask_param, n_freq) for _ in range(self.n_freq_masks)]
        if self.time_mask_param >= 1:
            masks += [(1,) + self._draw(self.time_mask_param, n_time) for _ in range(self.n_time_masks)]
        return masks

    def __call__(self, spectrogram: torch.Tensor) -> torch.Tensor:
        """(C, F, T) or (B, C, F, T) -> same shape, a new tensor when the augmentation fires (as ``masked_fill``)."""
        if random.random() > self.p:
            return spectrogram
        if spectrogram.dim() not in (3, 4):
            raise ValueError(f"SpecAugment: expected (C, F, T) or (B, C, F, T), got {tuple(spectrogram.shape)}")
        masks = self.draw_masks(spectrogram.shape[-2], spectrogram.shape[-1])
        if not masks:
            return spectrogram
        dev = _cuda_device()
        src = spectrogram.to(device=dev, dtype=torch.float32).contiguous()
        out = torch.empty_like(src)
        n = len(masks)
        arr = lambda k: (C.c_int * n)(*[m[k] for m in masks])
        n_img = src.numel() // (src.shape[-2] * src.shape[-1])
        if src.numel():
            _lib.check(_lib.load().cough_mask_axes(src.data_ptr(), out.data_ptr(), n_img, src.shape[-2], src.shape[-1], n,
                                                   arr(0), arr(1), arr(2), torch.cuda.current_stream(dev).cuda_stream),
                       "cough_mask_axes")
        return out.to(spectrogram.device) if spectrogram.device.type == "cpu" else out
