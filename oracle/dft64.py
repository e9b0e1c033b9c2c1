"""Independent float64 re-derivation of the feature chain (TEST INFRASTRUCTURE).

Deliberately shares no code with ``oracle/featurizer.py``: explicit reflect
padding and framing, ``numpy.fft.rfft`` in float64, analytic filterbank / DCT
formulas.  Used to cross-check the float32 torch restatement
(SURVEY.md section 8c "what pins the restatement instead").
"""
from __future__ import annotations

import numpy as np

SR, N_FFT, HOP, WIN, N_MELS, N_MFCC = 16000, 512, 160, 400, 64, 13


def window512(win: int = WIN, n_fft: int = N_FFT) -> np.ndarray:
    n = np.arange(win, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win)          # periodic Hann
    out = np.zeros(n_fft)
    left = (n_fft - win) // 2
    out[left:left + win] = w
    return out


def frames(x: np.ndarray, hop: int = HOP, n_fft: int = N_FFT) -> np.ndarray:
    """(N,) -> (T, n_fft) reflect-padded, T = 1 + N // hop."""
    x = np.asarray(x, dtype=np.float64)
    pad = n_fft // 2
    xp = np.concatenate([x[1:pad + 1][::-1], x, x[-pad - 1:-1][::-1]])
    t = 1 + (len(xp) - n_fft) // hop                         # = 1 + N // hop for an even n_fft
    idx = np.arange(n_fft)[None, :] + hop * np.arange(t)[:, None]
    return xp[idx]


def stft_power(x: np.ndarray, hop: int = HOP, win: int = WIN, n_fft: int = N_FFT) -> np.ndarray:
    """(N,) -> (n_fft // 2 + 1, T)."""
    f = frames(x, hop, n_fft) * window512(win, n_fft)[None, :]
    s = np.fft.rfft(f, axis=1)
    return (s.real ** 2 + s.imag ** 2).T


def mel_fb(f_min: float = 100.0, f_max: float = 4000.0, n_mels: int = N_MELS, sample_rate: int = SR,
           n_fft: int = N_FFT) -> np.ndarray:
    """(n_fft // 2 + 1, n_mels) HTK triangles, no area normalisation."""
    def hz2mel(f):
        return 2595.0 * np.log10(1.0 + f / 700.0)

    def mel2hz(m):
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)

    freqs = np.linspace(0.0, sample_rate // 2, n_fft // 2 + 1)
    pts = mel2hz(np.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2))
    fb = np.zeros((n_fft // 2 + 1, n_mels))
    for m in range(n_mels):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        rise = (freqs - lo) / (ce - lo)
        fall = (hi - freqs) / (hi - ce)
        fb[:, m] = np.maximum(0.0, np.minimum(rise, fall))
    return fb


def dct_ortho(n_mfcc: int = N_MFCC, n_mels: int = N_MELS) -> np.ndarray:
    """(n_mels, n_mfcc)."""
    n = np.arange(n_mels)[:, None] + 0.5
    k = np.arange(n_mfcc)[None, :]
    d = np.cos(np.pi / n_mels * n * k) * np.sqrt(2.0 / n_mels)
    d[:, 0] *= 1.0 / np.sqrt(2.0)
    return d


def features(x: np.ndarray, sample_rate: int = SR, n_mels: int = N_MELS, hop: int = HOP, win: int = WIN,
             f_min: float = 100.0, f_max: float = 4000.0, n_mfcc: int = N_MFCC, n_fft: int = N_FFT) -> np.ndarray:
    """(N,) float -> (n_mels + 2 n_mfcc, T) float64: mel rows, z-scored MFCC, delta.  Defaults: (16000,) -> (90, 101)."""
    mel = mel_fb(f_min, f_max, n_mels, sample_rate, n_fft).T @ stft_power(x, hop, win, n_fft)   # (n_mels, T)
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    mel_n = np.clip((db + 80.0) / 80.0, 0.0, 1.0)
    mfcc = dct_ortho(n_mfcc, n_mels).T @ db                     # (n_mfcc, T)
    z = (mfcc - mfcc.mean()) / (mfcc.std(ddof=1) + 1e-8)
    zp = np.concatenate([z[:, :1], z, z[:, -1:]], axis=1)
    delta = (zp[:, 2:] - zp[:, :-2]) / 2.0
    return np.concatenate([mel_n, z, delta], axis=0)


def resample_direct(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                    rolloff: float = 0.99) -> np.ndarray:
    """Band-limited interpolation by DIRECT evaluation of the published ``sinc_interp_hann`` formula (what
    ``torchaudio.functional.resample`` implements as a polyphase FIR), float64, no polyphase table, no strided
    convolution: output sample m sits at input position p = m * orig / new and

        y[m] = sum_n x[n] * (f_c / orig) * sinc(pi * tau) * cos^2(pi * tau / (2 * lpw)),   tau = f_c * (n - p) / orig,

    over the taps with |tau| < lpw (lpw = 6 zero crossings), f_c = rolloff * min(orig, new) in units where the rates
    are reduced by their gcd; samples outside [0, N) are zero; ceil(N * new / orig) outputs."""
    import math
    x = np.asarray(x, dtype=np.float64)
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    fc = min(orig, new) * rolloff
    n_out = int(math.ceil(len(x) * new / orig))
    half = lowpass_filter_width * orig / fc                      # the window's half-width in input samples
    y = np.zeros(n_out)
    for m in range(n_out):
        p = m * orig / new
        lo, hi = int(math.floor(p - half)) - 1, int(math.ceil(p + half)) + 1
        n = np.arange(max(lo, 0), min(hi, len(x) - 1) + 1)
        tau = fc * (n - p) / orig
        keep = np.abs(tau) < lowpass_filter_width
        tau, n = tau[keep], n[keep]
        s = np.where(tau == 0.0, 1.0, np.sin(np.pi * tau) / np.where(tau == 0.0, 1.0, np.pi * tau))
        y[m] = np.sum(x[n] * s * np.cos(np.pi * tau / (2 * lowpass_filter_width)) ** 2) * (fc / orig)
    return y


def spectral_centroid_direct(x: np.ndarray, sample_rate: int = SR) -> np.ndarray:
    """Centroid of the magnitude spectrogram taken with a periodic Hann(512) window (torchaudio's SpectralCentroid
    default window = n_fft), float64: sum_k f_k |X_k| / sum_k |X_k|, f_k = k * sr / 512.  (T,)"""
    n = np.arange(N_FFT, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)
    mag = np.abs(np.fft.rfft(frames(x) * w[None, :], axis=1))                 # (T, 257)
    f = np.arange(N_FFT // 2 + 1) * (sample_rate / N_FFT)
    return (mag * f[None, :]).sum(axis=1) / mag.sum(axis=1)
