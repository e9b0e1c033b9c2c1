"""torch-CPU oracle for the audio featuriser (TEST INFRASTRUCTURE, not product).

Restates ``/root/reference/src/preprocessing.py`` for the configuration the
reference ships (``/root/reference/src/train.py:264-287``): log-mel(64) +
MFCC(13) + delta-MFCC(13) -> (1, 90, 101), plus the two cheap optional flags
(pre-emphasis, delta-delta), the PCEN branch, the spectral-contrast / centroid rows
(``extract_spectral_contrast``) and the front of ``process`` (resample, mono, pad / trim).

PARITY UNPINNED against torchaudio: the reference delegates STFT / mel / dB /
DCT to ``torchaudio.transforms`` (``preprocessing.py:94-127``), which is not
available in this environment (see ``oracle/__init__.py``).  The whole shipped
chain is, however, cross-checked end to end against code the builder did not
write (``transformers.audio_utils.spectrogram`` + ``scipy.fft.dct``: 7e-6 on the
32 golden clips, ``tests/test_oracle_featurizer.py``).  The functions
below restate torchaudio's published algorithms with the same float32 torch
op sequence so that the tables are built the way torchaudio builds them:

=====================  =======================================================
here                   reference call site -> torchaudio algorithm restated
=====================  =======================================================
``hann_window``        preprocessing.py:94-106 -> Spectrogram(window_fn=torch.hann_window)
``stft_power``         preprocessing.py:398,425 -> functional.spectrogram(power=2,
                       center=True, pad_mode="reflect", normalized=False, onesided=True)
``melscale_fbanks``    MelScale(n_mels, sr, f_min, f_max, n_stft, norm=None, "htk")
``amplitude_to_db``    preprocessing.py:109-112,407 -> functional.amplitude_to_DB
                       (multiplier=10, amin=1e-10, db_multiplier=0, top_db=80, per clip)
``create_dct``         preprocessing.py:116-127 -> functional.create_dct(13, 64, "ortho")
=====================  =======================================================
"""
from __future__ import annotations

import math
import warnings
from typing import List, Optional

import numpy as np
import torch

SAMPLE_RATE = 16000
N_FFT = 512
HOP = 160
WIN = 400
N_MELS = 64
N_MFCC = 13
F_MIN = 100.0
F_MAX = 4000.0
TOP_DB = 80.0
AMIN = 1e-10


# --------------------------------------------------------------------------- tables
def hann_window(win_length: int = WIN) -> torch.Tensor:
    """Periodic Hann, as ``torch.hann_window`` (torchaudio's default window_fn)."""
    return torch.hann_window(win_length, periodic=True, dtype=torch.float32)


def hz_to_mel_htk(freq: float) -> float:
    return 2595.0 * math.log10(1.0 + (freq / 700.0))


def mel_to_hz_htk(mels: torch.Tensor) -> torch.Tensor:
    return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)


def melscale_fbanks(n_freqs: int = N_FFT // 2 + 1, f_min: float = F_MIN, f_max: float = F_MAX,
                    n_mels: int = N_MELS, sample_rate: int = SAMPLE_RATE) -> torch.Tensor:
    """(n_freqs, n_mels) triangular HTK filterbank, norm=None, float32 op order of torchaudio."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = mel_to_hz_htk(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def create_dct(n_mfcc: int = N_MFCC, n_mels: int = N_MELS) -> torch.Tensor:
    """(n_mels, n_mfcc) DCT-II, norm='ortho'."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t().contiguous()


# --------------------------------------------------------------------------- transforms
def stft_power(waveform: torch.Tensor, n_fft: int = N_FFT, hop: int = HOP, win: int = WIN,
               power: float = 2.0) -> torch.Tensor:
    """(..., N) -> (..., n_fft//2+1, 1+N//hop) spectrogram (F1): torchaudio.functional.spectrogram with
    pad=0, normalized=False, center=True, reflect padding; ``power=2`` is |X|^2, ``power=1`` is |X|."""
    shape = waveform.shape
    w = waveform.reshape(-1, shape[-1])
    spec = torch.stft(w, n_fft=n_fft, hop_length=hop, win_length=win, window=hann_window(win),
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    spec = spec.reshape(shape[:-1] + spec.shape[-2:])
    return spec.abs() if power == 1.0 else spec.abs().pow(power)


def mel_spectrogram(waveform: torch.Tensor, fb: Optional[torch.Tensor] = None, n_fft: int = N_FFT, hop: int = HOP,
                    win: int = WIN) -> torch.Tensor:
    """F1+F2: (..., N) -> (..., n_mels, T)."""
    fb = melscale_fbanks() if fb is None else fb
    spec = stft_power(waveform, n_fft=n_fft, hop=hop, win=win)
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


def amplitude_to_db(x: torch.Tensor, top_db: Optional[float] = TOP_DB) -> torch.Tensor:
    """F3.  The top_db floor is taken per packed clip exactly as torchaudio does:
    for >2-D input dim -3 is the channel dim and the max runs over (C, F, T)."""
    x_db = 10.0 * torch.log10(torch.clamp(x, min=AMIN))
    x_db = x_db - 10.0 * math.log10(max(AMIN, 1.0))
    if top_db is not None:
        shape = x_db.size()
        packed = shape[-3] if x_db.dim() > 2 else 1
        x_db = x_db.reshape(-1, packed, shape[-2], shape[-1])
        x_db = torch.max(x_db, (x_db.amax(dim=(-3, -2, -1)) - top_db).view(-1, 1, 1, 1))
        x_db = x_db.reshape(shape)
    return x_db


def mfcc_transform(waveform: torch.Tensor, fb=None, dct=None, **stft) -> torch.Tensor:
    """F5 (un-normalised): second, parameter-identical STFT->mel->dB chain, then DCT."""
    dct = create_dct() if dct is None else dct
    mel_db = amplitude_to_db(mel_spectrogram(waveform, fb, **stft))
    return torch.matmul(mel_db.transpose(-1, -2), dct).transpose(-1, -2)


# --------------------------------------------------------------------------- reference methods
def normalize(waveform: torch.Tensor) -> torch.Tensor:
    """F0 -- preprocessing.py:199-212."""
    max_val = waveform.abs().max()
    if max_val > 0:
        return waveform / max_val
    return waveform


def pre_emphasis(waveform: torch.Tensor, coef: float = 0.97) -> torch.Tensor:
    """preprocessing.py:214-240 (first sample kept)."""
    return torch.cat([waveform[:, :1], waveform[:, 1:] - coef * waveform[:, :-1]], dim=1)


def compute_deltas(features: torch.Tensor) -> torch.Tensor:
    """F7 -- preprocessing.py:342-356 (replicate pad, central difference / 2)."""
    padded = torch.nn.functional.pad(features, (1, 1), mode="replicate")
    return (padded[:, :, 2:] - padded[:, :, :-2]) / 2


def pad_or_trim(waveform: torch.Tensor, length: int = SAMPLE_RATE) -> torch.Tensor:
    """preprocessing.py:358-385 (centre trim / centred zero pad)."""
    cur = waveform.shape[1]
    if cur == length:
        return waveform
    if cur > length:
        start = (cur - length) // 2
        return waveform[:, start:start + length]
    padding = length - cur
    left = padding // 2
    return torch.nn.functional.pad(waveform, (left, padding - left))


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional.resample's kernel (sinc_interp_hann), built in float64 and cast to float32."""
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
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels *= window * (base_freq / orig)
    return kernels.to(torch.float32), width, orig, new


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int = SAMPLE_RATE) -> torch.Tensor:
    """T.Resample(orig_freq, new_freq)(waveform) -- preprocessing.py:146-183: zero-pad (width, width + orig),
    conv1d with the (new, 1, K) kernel at stride orig, interleave the phases, cut to ceil(new * N / orig)."""
    if orig_freq == new_freq:
        return waveform
    kernel, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    shape = waveform.shape
    w = waveform.reshape(-1, shape[-1])
    length = w.shape[1]
    w = torch.nn.functional.pad(w, (width, width + orig))
    res = torch.nn.functional.conv1d(w[:, None], kernel, stride=orig)
    res = res.transpose(1, 2).reshape(w.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return res[..., :target].reshape(shape[:-1] + (target,))


def to_mono(waveform: torch.Tensor) -> torch.Tensor:
    """preprocessing.py:185-197."""
    return waveform if waveform.shape[0] == 1 else waveform.mean(dim=0, keepdim=True)


def process(waveform: torch.Tensor, orig_sr: int, **feature_kw) -> torch.Tensor:
    """preprocessing.py:491-517: resample -> mono -> normalize -> pad/trim -> extract_features."""
    w = pad_or_trim(normalize(to_mono(resample(waveform, orig_sr))))
    return extract_features(w, **feature_kw)


def apply_pcen(mel_spec: torch.Tensor, alpha: float = 0.98, delta: float = 2.0, r: float = 0.5,
               eps: float = 1e-6) -> torch.Tensor:
    """preprocessing.py:305-340: moving average over 10 frames (zero padded, count_include_pad), then
    (mel / (eps + smooth)^alpha + delta)^r - delta^r."""
    smooth = torch.nn.functional.avg_pool2d(mel_spec.unsqueeze(0), kernel_size=(1, 10), stride=(1, 1),
                                            padding=(0, 5)).squeeze(0)
    smooth = smooth[:, :, :mel_spec.shape[2]]
    return (mel_spec / (eps + smooth).pow(alpha) + delta).pow(r) - delta ** r


def extract_mel_spectrogram(waveform: torch.Tensor, fb=None, use_pcen: bool = False, **stft) -> torch.Tensor:
    """F1-F4 -- preprocessing.py:387-412 (PCEN branch :400-404, log branch :405-410)."""
    if use_pcen:
        m = apply_pcen(mel_spectrogram(waveform, fb, **stft))
        return (m - m.min()) / (m.max() - m.min() + 1e-8)
    mel_db = amplitude_to_db(mel_spectrogram(waveform, fb, **stft))
    return ((mel_db + 80) / 80).clamp(0, 1)


def extract_mfcc(waveform: torch.Tensor, fb=None, dct=None, **stft) -> torch.Tensor:
    """F5-F6 -- preprocessing.py:414-430 (global z-score, unbiased std)."""
    mfcc = mfcc_transform(waveform, fb, dct, **stft)
    return (mfcc - mfcc.mean()) / (mfcc.std() + 1e-8)


def contrast_band_edges(n_bands: int = 6, n_freq: int = N_FFT // 2 + 1) -> List[int]:
    """preprocessing.py:267-268: ``torch.logspace(0, log10(n_freq), n_bands + 2).int()`` clamped to [0, n_freq]."""
    edges = torch.logspace(0, float(np.log10(n_freq)), n_bands + 2).int()
    return torch.clamp(edges, 0, n_freq).tolist()


def spectral_centroid(waveform: torch.Tensor, sample_rate: int = SAMPLE_RATE, n_fft: int = N_FFT,
                      hop: int = HOP) -> torch.Tensor:
    """torchaudio.functional.spectral_centroid as T.SpectralCentroid(sample_rate, n_fft, hop_length) calls it
    (preprocessing.py:137-141): win_length defaults to n_fft (Hann(512), not the featuriser's Hann(400)), pad 0,
    magnitude spectrogram; ``(freqs * |X|).sum(freq) / |X|.sum(freq)`` with freqs = linspace(0, sr // 2, 257)."""
    spec = stft_power(waveform, n_fft=n_fft, hop=hop, win=n_fft, power=1.0)          # (..., 257, T)
    freqs = torch.linspace(0, sample_rate // 2, steps=1 + n_fft // 2).reshape(-1, 1)
    return (freqs * spec).sum(dim=-2) / spec.sum(dim=-2)


def extract_spectral_contrast(waveform: torch.Tensor, n_bands: int = 6, sample_rate: int = SAMPLE_RATE,
                              n_fft: int = N_FFT, hop: int = HOP, win: int = WIN) -> torch.Tensor:
    """preprocessing.py:242-303, statement by statement: (1, N) -> (1, n_bands + 1, T).

    With the default n_bands = 6 (and any n_bands >= 5) the first band is the single bin [1, 2): ``top_idx =
    max(1, int(1 * 0.8)) = 1`` makes ``sorted_band[:, 1:, :]`` empty, its mean is NaN, and the global z-score at
    the end spreads the NaN over every row.  The restatement keeps that behaviour."""
    spec = stft_power(waveform, n_fft=n_fft, hop=hop, win=win)                        # T.Spectrogram(power=2.0)
    n_freq, n_time = spec.shape[1], spec.shape[2]
    band_edges = contrast_band_edges(n_bands, n_freq)
    contrast = torch.zeros(1, n_bands + 1, n_time)
    for i in range(n_bands):
        low, high = band_edges[i], band_edges[i + 1]
        if high <= low:
            high = low + 1
        if high > n_freq:
            high = n_freq
        band = spec[:, low:high, :]
        if band.shape[1] > 0:
            sorted_band, _ = torch.sort(band, dim=1)
            n_bins = sorted_band.shape[1]
            top_idx = max(1, int(n_bins * 0.8))
            bot_idx = max(1, int(n_bins * 0.2))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")                                        # mean of an empty slice
                peaks = sorted_band[:, top_idx:, :].mean(dim=1)
            valleys = sorted_band[:, :bot_idx, :].mean(dim=1)
            contrast[:, i, :] = torch.log1p(peaks) - torch.log1p(valleys)
    centroid = spectral_centroid(waveform, sample_rate, n_fft=n_fft, hop=hop) / (sample_rate / 2)
    contrast[:, -1, :centroid.shape[1]] = centroid
    return (contrast - contrast.mean()) / (contrast.std() + 1e-8)


def extract_features(waveform: torch.Tensor, use_pre_emphasis: bool = False, pre_emphasis_coef: float = 0.97,
                     use_delta_delta: bool = False, use_pcen: bool = False, fb=None, dct=None,
                     use_mfcc: bool = True, use_spectral_contrast: bool = False,
                     n_contrast_bands: int = 6, sample_rate: int = SAMPLE_RATE, n_fft: int = N_FFT, hop: int = HOP,
                     win: int = WIN) -> torch.Tensor:
    """F8 -- preprocessing.py:432-489: (1, N) -> (1, 90 [or 103, +7 with spectral contrast], T).  ``fb`` / ``dct`` are the
    tables of the constructor's n_mels / n_mfcc / f_min / f_max (``melscale_fbanks`` / ``create_dct``); ``sample_rate`` /
    ``n_fft`` / ``hop`` / ``win`` its STFT geometry (:94-141)."""
    stft = dict(n_fft=n_fft, hop=hop, win=win)
    w = pre_emphasis(waveform, pre_emphasis_coef) if use_pre_emphasis else waveform
    mel = extract_mel_spectrogram(w, fb, use_pcen, **stft)
    feats = [mel]
    if use_mfcc:
        mfcc = extract_mfcc(w, fb, dct, **stft)
        delta = compute_deltas(mfcc)
        feats += [mfcc, delta]
        if use_delta_delta:
            feats.append(compute_deltas(delta))
    if use_spectral_contrast:   # from the un-emphasised signal (:476-478)
        feats.append(extract_spectral_contrast(waveform, n_contrast_bands, sample_rate, **stft))
    t = min(f.shape[2] for f in feats)
    return torch.cat([f[:, :, :t] for f in feats], dim=1)


def geometry_kwargs(sample_rate: int = SAMPLE_RATE, n_mels: int = N_MELS, n_fft: int = N_FFT, hop_length: int = HOP,
                    win_length: int = WIN, f_min: float = F_MIN, f_max: float = F_MAX, n_mfcc: int = N_MFCC) -> dict:
    """``extract_features`` keyword arguments (tables + STFT geometry) of an ``AudioPreprocessor(...)`` constructor call
    (preprocessing.py:32-51, :94-141)."""
    return dict(fb=melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate), dct=create_dct(n_mfcc, n_mels),
                sample_rate=sample_rate, n_fft=n_fft, hop=hop_length, win=win_length)


def extract_features_batch(waveforms: torch.Tensor, normalize_first: bool = False, **kw) -> torch.Tensor:
    """(B, N) -> (B, F, T) by looping the per-clip reference path (per-clip reductions)."""
    out = []
    kw.setdefault("fb", melscale_fbanks())
    kw.setdefault("dct", create_dct())
    for i in range(waveforms.shape[0]):
        w = waveforms[i:i + 1]
        if normalize_first:
            w = normalize(w)
        out.append(extract_features(w, **kw))
    return torch.cat(out, dim=0)


def extract_features_batched_fast(waveforms: torch.Tensor, normalize_first: bool = False) -> torch.Tensor:
    """Best-effort batched CPU path for the cpu_baseline leg: one STFT for both
    branches, whole batch at once, per-clip reductions kept (shipped flags only)."""
    w = waveforms
    if normalize_first:
        m = w.abs().amax(dim=1, keepdim=True)
        w = torch.where(m > 0, w / torch.where(m > 0, m, torch.ones_like(m)), w)
    mel = mel_spectrogram(w)                                   # (B, 64, T)
    db = 10.0 * torch.log10(torch.clamp(mel, min=AMIN))
    db = torch.max(db, db.amax(dim=(1, 2), keepdim=True) - TOP_DB)
    mel_n = ((db + 80) / 80).clamp(0, 1)
    mfcc = torch.matmul(db.transpose(1, 2), create_dct()).transpose(1, 2)
    mean = mfcc.mean(dim=(1, 2), keepdim=True)
    std = mfcc.reshape(mfcc.shape[0], -1).std(dim=1).view(-1, 1, 1)
    z = (mfcc - mean) / (std + 1e-8)
    return torch.cat([mel_n, z, compute_deltas(z)], dim=1)


class RealtimeWindowerOracle:
    """R0 -- RealtimePreprocessor.add_audio/reset (preprocessing.py:553-616)."""

    def __init__(self, window_duration: float = 1.0, hop_duration: float = 0.5,
                 sample_rate: int = SAMPLE_RATE, **feature_kw):
        self.window_samples = int(sample_rate * window_duration)
        self.hop_samples = int(sample_rate * hop_duration)
        self.feature_kw = feature_kw
        self.buffer = torch.zeros(1, 0)

    def add_audio(self, chunk: torch.Tensor) -> List[torch.Tensor]:
        if chunk.dim() == 1:
            chunk = chunk.unsqueeze(0)
        self.buffer = torch.cat([self.buffer, chunk], dim=1)
        out = []
        while self.buffer.shape[1] >= self.window_samples:
            window = normalize(self.buffer[:, :self.window_samples])
            out.append(extract_features(window, **self.feature_kw))
            self.buffer = self.buffer[:, self.hop_samples:]
        return out

    def reset(self):
        self.buffer = torch.zeros(1, 0)
