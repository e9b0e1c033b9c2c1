"""Immutable float32 tables handed to the HIP featuriser at create time.

Built on the host with the float32 torch op sequence torchaudio uses for
``T.MelSpectrogram`` / ``T.MFCC`` (the transforms the reference constructs at
``/root/reference/src/preprocessing.py:94-127``), so the device kernels see the
same table values the reference's CPU path multiplies by.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def hann_window(win_length: int) -> torch.Tensor:
    return torch.hann_window(win_length, periodic=True, dtype=torch.float32)


def mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """(n_freqs, n_mels); HTK mel scale, no area normalisation."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down_slopes, up_slopes)).contiguous()


def dct_matrix(n_mfcc: int, n_mels: int) -> torch.Tensor:
    """(n_mels, n_mfcc); DCT-II with orthonormal scaling."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t().contiguous()


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Polyphase windowed-sinc kernel of ``T.Resample(orig_freq, new_freq)`` (sinc_interp_hann, the reference
    constructs it with defaults at ``/root/reference/src/preprocessing.py:146-153``): (new, 2*width + orig)
    float32 computed in float64 as torchaudio does, plus ``width``.  Frequencies are reduced by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32).reshape(new, 2 * width + orig).contiguous(), width, orig, new


def contrast_band_edges(n_bands: int, n_freq: int) -> list:
    """Band edges of extract_spectral_contrast, computed with the reference's own expression
    (src/preprocessing.py:267-268): ``torch.logspace(0, np.log10(n_freq), n_bands + 2).int()`` clamped to
    [0, n_freq].  n_bands + 2 integers; the loop uses the first n_bands + 1."""
    edges = torch.logspace(0, float(np.log10(n_freq)), n_bands + 2).int()
    return torch.clamp(edges, 0, n_freq).tolist()
