"""CPU restatement of the sliding-window detection loop (TEST INFRASTRUCTURE).

Follows ``/root/reference/src/inference.py``: ``predict`` (:165-189),
``process_audio_chunk`` (:191-241), ``reset`` (:243-247), state (:110-117).
The reference reads ``datetime.now()`` (:226,233); here the clock is injected
so results are deterministic.  Composes the other oracles: windows come from
``featurizer.RealtimeWindowerOracle`` (hop 0.25 s, inference.py:98), the
probability from ``resnet.forward``.
"""
from __future__ import annotations

from collections import deque
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch

from . import featurizer, resnet


class EngineOracle:
    def __init__(self, state_dict: Dict[str, torch.Tensor], confidence_threshold: float = 0.5,
                 smoothing_window: int = 3, debounce_seconds: float = 0.5,
                 clock: Optional[Callable[[], float]] = None):
        self.sd = state_dict
        self.confidence_threshold = confidence_threshold
        self.debounce_seconds = debounce_seconds
        self.windower = featurizer.RealtimeWindowerOracle(window_duration=1.0, hop_duration=0.25)
        self.history = deque(maxlen=smoothing_window)
        self.last_detection_time = 0.0
        self.clock = clock or (lambda: 0.0)
        self.window_probs = []          # every per-window probability, for parity checks

    def predict(self, spec: torch.Tensor) -> Tuple[bool, float]:
        if spec.dim() == 3:
            spec = spec.unsqueeze(0)
        probs = torch.softmax(resnet.forward(spec, self.sd), dim=1)
        p = probs[0, 1].item()
        return p > 0.5, p

    def process_audio_chunk(self, chunk) -> Optional[Tuple[float, float]]:
        if isinstance(chunk, np.ndarray):
            chunk = torch.from_numpy(chunk.astype(np.float32))
        if chunk.dim() == 1:
            chunk = chunk.unsqueeze(0)
        if chunk.shape[0] > 1:
            chunk = chunk.mean(dim=0, keepdim=True)
        for spec in self.windower.add_audio(chunk):
            _, conf = self.predict(spec)
            self.window_probs.append(conf)
            self.history.append(conf)
            smoothed = float(np.mean(self.history))
            now = self.clock()
            if smoothed >= self.confidence_threshold and now - self.last_detection_time >= self.debounce_seconds:
                self.last_detection_time = now
                return now, smoothed        # remaining windows of this chunk are dropped (inference.py:239)
        return None

    def reset(self):
        self.windower.reset()
        self.history.clear()
        self.last_detection_time = 0.0
