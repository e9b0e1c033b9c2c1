"""Seeded synthetic 1 s @ 16 kHz clips (bench / test inputs; no dataset is available offline).

The value distribution follows the reference's own synthetic generators
(``/root/reference/setup_coughvid.py:381-441`` cough-like burst / silence /
white noise / hum / clicks / speech-like; ``/root/reference/prepare_data.py:136-163``)
shortened to 1 s, mixture selected by ``seed % 6``.  Clip
``i`` depends only on ``i`` so any rank can regenerate its own shard.
"""
from __future__ import annotations

import numpy as np
import torch

SAMPLE_RATE = 16000
N = 16000
KINDS = ("cough", "silence", "white", "hum", "clicks", "speech")


def make_clip(seed: int, peak_normalize: bool = True) -> np.ndarray:
    """One float32 clip of 16000 samples, deterministic in ``seed``."""
    rng = np.random.default_rng(int(seed))
    t = np.arange(N, dtype=np.float64) / SAMPLE_RATE
    kind = int(seed) % 6
    if kind == 0:       # cough-like burst over a noise floor
        dur = rng.uniform(0.3, 0.8)
        n_burst = int(dur * SAMPLE_RATE)
        start = int(rng.uniform(0.0, 1.0 - dur) * SAMPLE_RATE)
        n_att = int(0.02 * SAMPLE_RATE)
        env_b = np.concatenate([np.linspace(0.0, 1.0, n_att), np.exp(-np.linspace(0.0, 5.0, n_burst - n_att))])
        env = np.zeros(N)
        env[start:start + n_burst] = env_b[:N - start]
        body = (0.7 * rng.standard_normal(N)
                + 0.2 * np.sin(2 * np.pi * rng.uniform(80, 150) * t)
                + 0.1 * np.sin(2 * np.pi * rng.uniform(200, 400) * t))
        x = env * body
        x = x / (np.abs(x).max() + 1e-8) * 0.8 + rng.standard_normal(N) * 0.01
    elif kind == 1:     # near silence
        x = rng.standard_normal(N) * 0.005
    elif kind == 2:     # white noise
        x = rng.standard_normal(N) * rng.uniform(0.02, 0.1)
    elif kind == 3:     # mains-like hum
        x = np.sin(2 * np.pi * rng.choice([50, 60, 100, 120]) * t) * 0.1 + rng.standard_normal(N) * 0.02
    elif kind == 4:     # clicks on a floor
        x = rng.standard_normal(N) * 0.01
        for _ in range(int(rng.integers(1, 5))):
            p = int(rng.integers(0, N - 100))
            x[p:p + 50] = rng.uniform(-0.3, 0.3)
    else:               # speech-like formant stack
        x = np.zeros(N)
        for _ in range(int(rng.integers(2, 5))):
            x += np.sin(2 * np.pi * rng.uniform(100, 1000) * t) * rng.uniform(0.05, 0.15)
        x += rng.standard_normal(N) * 0.02
    x = x.astype(np.float32)
    if peak_normalize:
        m = np.abs(x).max()
        if m > 0:
            x = x / m
    return x


def make_clips(start: int, count: int, stride: int = 1, peak_normalize: bool = True) -> np.ndarray:
    """(count, 16000) float32: clips ``start, start+stride, ...`` (round-robin shards use stride=W)."""
    out = np.empty((count, N), dtype=np.float32)
    for j in range(count):
        out[j] = make_clip(start + j * stride, peak_normalize)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Counter-based variant of the same recipe: every sample is a pure function of (seed, sample index), so the clip
# stream can be generated ON THE DEVICE by any rank (csrc/synth.hip: cough_synth_clips, BASELINE.json configs[3]).
# This is the host mirror of that kernel -- the same float32 operation sequence -- used to check it sample by sample.
_M32 = np.uint64(0xFFFFFFFF)
_F = np.float32
_TWO_PI = _F(6.28318530717958647692)
_U24 = _F(5.9604644775390625e-8)


def _hash32(x):
    x = np.asarray(x, dtype=np.uint64) & _M32
    x = x ^ (x >> np.uint64(16))
    x = (x * np.uint64(0x7feb352d)) & _M32
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x846ca68b)) & _M32
    return x ^ (x >> np.uint64(16))


def _key(seed: int, stream: int, idx):
    base = _hash32((np.uint64(seed) * np.uint64(0x9E3779B9) + np.uint64(stream)) & _M32)
    return _hash32((base + np.asarray(idx, dtype=np.uint64)) & _M32)


def _u01(k):
    return (k >> np.uint64(8)).astype(_F) * _U24


def _normal(seed: int, stream: int, i):
    i = np.asarray(i, dtype=np.uint64)
    u1 = ((_key(seed, stream, 2 * i) >> np.uint64(8)) + np.uint64(1)).astype(_F) * _U24
    u2 = _u01(_key(seed, stream, 2 * i + 1))
    return np.sqrt(_F(-2.0) * np.log(u1)) * np.cos(_TWO_PI * u2)


def _param(seed: int, p: int):
    return _F(_u01(_key(seed, 0xF00D, p)))


def _tone(fq: int, i):
    """sin(2 pi f t) for f = fq/16 Hz: phase fq*i/256000 reduced in exact integer arithmetic."""
    r = (np.uint64(fq) * i.astype(np.uint64)) % np.uint64(256000)
    return np.sin(_TWO_PI * (r.astype(_F) * _F(3.90625e-6)))


def _freq16(lo16: float, span16: float, x) -> int:
    return int(_F(lo16) + _F(span16) * x)


def make_clip_counter(seed: int) -> np.ndarray:
    """Host mirror of ``cough_synth_clips`` (csrc/synth.hip): one un-normalised float32 clip of 16000 samples."""
    seed64 = int(seed)
    s = seed64 & 0xFFFFFFFF
    kind = seed64 % 6
    i = np.arange(N)
    if kind == 0:
        dur = _F(0.3) + _F(0.5) * _param(s, 0)
        n_burst = int(dur * _F(16000.0))
        start = int((_param(s, 1) * (_F(1.0) - dur)) * _F(16000.0))
        n_att = 320
        f1 = _freq16(1280.0, 1120.0, _param(s, 2))           # 80-150 Hz in 1/16 Hz
        f2 = _freq16(3200.0, 3200.0, _param(s, 3))           # 200-400 Hz
        inv_att, inv_dec = _F(1.0) / _F(n_att - 1), _F(5.0) / _F(n_burst - n_att - 1)
        j = i - start
        inside = (j >= 0) & (j < n_burst)
        env = np.where(j < n_att, j.astype(_F) * inv_att, np.exp(-((j - n_att).astype(_F) * inv_dec)))
        body = _F(0.7) * _normal(s, 1, i) + _F(0.2) * _tone(f1, i) + _F(0.1) * _tone(f2, i)
        x = np.where(inside, env * body, _F(0.0)).astype(_F)
        g = _F(0.8) / (np.abs(x).max() + _F(1e-8))
        x = x * g + _F(0.01) * _normal(s, 2, i)
    elif kind == 1:
        x = _F(0.005) * _normal(s, 1, i)
    elif kind == 2:
        x = (_F(0.02) + _F(0.08) * _param(s, 0)) * _normal(s, 1, i)
    elif kind == 3:
        f = (800, 960, 1600, 1920)[int(_param(s, 0) * _F(4.0))]        # 50 / 60 / 100 / 120 Hz
        x = _F(0.1) * _tone(f, i) + _F(0.02) * _normal(s, 1, i)
    elif kind == 4:
        x = _F(0.01) * _normal(s, 1, i)
        for c in range(1 + int(_param(s, 0) * _F(4.0))):
            p = int(_param(s, 1 + 2 * c) * _F(N - 100))
            x[p:p + 50] = _F(-0.3) + _F(0.6) * _param(s, 2 + 2 * c)
    else:
        x = np.zeros(N, dtype=_F)
        for c in range(2 + int(_param(s, 0) * _F(3.0))):
            f = _freq16(1600.0, 14400.0, _param(s, 1 + 2 * c))   # 100-1000 Hz
            a = _F(0.05) + _F(0.1) * _param(s, 2 + 2 * c)
            x = x + a * _tone(f, i)
        x = x + _F(0.02) * _normal(s, 1, i)
    return np.asarray(x, dtype=_F)


def device_clips(first_seed: int, count: int, seed_stride: int = 1, device=None, out: "torch.Tensor" = None):
    """(count, 16000) float32 CUDA tensor of synthetic clips generated on the device by ``cough_synth_clips``
    (clip j has seed ``first_seed + j*seed_stride``; un-normalised level)."""
    from . import _lib
    if not torch.cuda.is_available():
        raise RuntimeError("cough_detector_amd needs an AMD GPU (gfx950); there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if out is None:
        out = torch.empty((count, N), dtype=torch.float32, device=dev)
    if tuple(out.shape) != (count, N) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError("out must be a (count, 16000) float32 tensor with unit inner stride")
    if count == 0:
        return out
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().cough_synth_clips(out.data_ptr(), out.stride(0) if count > 1 else N, count, int(first_seed),
                                                 int(seed_stride), torch.cuda.current_stream(out.device).cuda_stream),
                   "cough_synth_clips")
    return out


def make_stream(seed: int, seconds: float) -> np.ndarray:
    """A mic-like stream: consecutive synthetic clips at un-normalised level."""
    n = int(round(seconds * SAMPLE_RATE))
    parts = [make_clip(seed * 1000 + k, peak_normalize=False) for k in range((n + N - 1) // N)]
    return np.concatenate(parts)[:n]


def random_state_dict(seed: int = 0, channels=(32, 64, 128)) -> dict:
    """Random-init classifier weights with the reference's state_dict key set / shapes
    (``/root/reference/src/model.py:216-283``) and NON-trivial BatchNorm statistics (fresh-init 0/1
    stats would not exercise BN folding).  No checkpoint is available offline."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k):
        bound = 1.0 / (ci * k * k) ** 0.5
        sd[name + ".weight"] = (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=g) + 0.5
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.2
        sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.2
        sd[name + ".running_var"] = torch.rand(c, generator=g) * 1.5 + 0.25
        sd[name + ".num_batches_tracked"] = torch.tensor(100)

    conv("conv1.0", channels[0], 1, 7)
    bn("conv1.1", channels[0])
    ci = channels[0]
    for i, co in enumerate(channels[1:]):
        p = f"res_blocks.{i}"
        conv(p + ".conv1", co, ci, 3)
        bn(p + ".bn1", co)
        conv(p + ".conv2", co, co, 3)
        bn(p + ".bn2", co)
        conv(p + ".skip.0", co, ci, 1)
        bn(p + ".skip.1", co)
        ci = co
    sd["fc.2.weight"] = (torch.rand(2, ci, generator=g) * 2 - 1) / ci ** 0.5
    sd["fc.2.bias"] = (torch.rand(2, generator=g) * 2 - 1) / ci ** 0.5
    return sd
