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


def make_stream(seed: int, seconds: float) -> np.ndarray:
    """A mic-like stream: consecutive synthetic clips at un-normalised level."""
    n = int(round(seconds * SAMPLE_RATE))
    parts = [make_clip(seed * 1000 + k, peak_normalize=False) for k in range((n + N - 1) // N)]
    return np.concatenate(parts)[:n]


def random_state_dict(seed: int = 0) -> dict:
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

    conv("conv1.0", 32, 1, 7)
    bn("conv1.1", 32)
    ci = 32
    for i, co in enumerate((64, 128)):
        p = f"res_blocks.{i}"
        conv(p + ".conv1", co, ci, 3)
        bn(p + ".bn1", co)
        conv(p + ".conv2", co, co, 3)
        bn(p + ".bn2", co)
        conv(p + ".skip.0", co, ci, 1)
        bn(p + ".skip.1", co)
        ci = co
    sd["fc.2.weight"] = (torch.rand(2, 128, generator=g) * 2 - 1) / 128 ** 0.5
    sd["fc.2.bias"] = (torch.rand(2, generator=g) * 2 - 1) / 128 ** 0.5
    return sd
