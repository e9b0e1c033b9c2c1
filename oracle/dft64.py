"""Independent float64 re-derivation of the feature chain (TEST INFRASTRUCTURE).

Deliberately shares no code with ``oracle/featurizer.py``: explicit reflect
padding and framing, ``numpy.fft.rfft`` in float64, analytic filterbank / DCT
formulas.  Used to cross-check the float32 torch restatement
(SURVEY.md section 8c "what pins the restatement instead").
"""
from __future__ import annotations

import numpy as np

SR, N_FFT, HOP, WIN, N_MELS, N_MFCC = 16000, 512, 160, 400, 64, 13


def window512() -> np.ndarray:
    n = np.arange(WIN, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / WIN)          # periodic Hann
    out = np.zeros(N_FFT)
    left = (N_FFT - WIN) // 2
    out[left:left + WIN] = w
    return out


def frames(x: np.ndarray) -> np.ndarray:
    """(N,) -> (T, 512) reflect-padded, hop 160, T = 1 + N // 160."""
    x = np.asarray(x, dtype=np.float64)
    pad = N_FFT // 2
    xp = np.concatenate([x[1:pad + 1][::-1], x, x[-pad - 1:-1][::-1]])
    t = 1 + len(x) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(t)[:, None]
    return xp[idx]


def stft_power(x: np.ndarray) -> np.ndarray:
    """(N,) -> (257, T)."""
    f = frames(x) * window512()[None, :]
    s = np.fft.rfft(f, axis=1)
    return (s.real ** 2 + s.imag ** 2).T


def mel_fb(f_min: float = 100.0, f_max: float = 4000.0) -> np.ndarray:
    """(257, 64) HTK triangles, no area normalisation."""
    def hz2mel(f):
        return 2595.0 * np.log10(1.0 + f / 700.0)

    def mel2hz(m):
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)

    freqs = np.linspace(0.0, SR / 2, N_FFT // 2 + 1)
    pts = mel2hz(np.linspace(hz2mel(f_min), hz2mel(f_max), N_MELS + 2))
    fb = np.zeros((N_FFT // 2 + 1, N_MELS))
    for m in range(N_MELS):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        rise = (freqs - lo) / (ce - lo)
        fall = (hi - freqs) / (hi - ce)
        fb[:, m] = np.maximum(0.0, np.minimum(rise, fall))
    return fb


def dct_ortho() -> np.ndarray:
    """(64, 13)."""
    n = np.arange(N_MELS)[:, None] + 0.5
    k = np.arange(N_MFCC)[None, :]
    d = np.cos(np.pi / N_MELS * n * k) * np.sqrt(2.0 / N_MELS)
    d[:, 0] *= 1.0 / np.sqrt(2.0)
    return d


def features(x: np.ndarray) -> np.ndarray:
    """(16000,) float -> (90, 101) float64: mel[0:64], z-scored MFCC[64:77], delta[77:90]."""
    mel = mel_fb().T @ stft_power(x)                            # (64, T)
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    mel_n = np.clip((db + 80.0) / 80.0, 0.0, 1.0)
    mfcc = dct_ortho().T @ db                                   # (13, T)
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
