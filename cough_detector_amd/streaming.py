"""Many concurrent audio streams on one GPU (BASELINE.json configs[4]).

The reference serves ONE stream: ``RealtimePreprocessor.add_audio`` keeps a host FIFO, cuts 1 s windows
every ``hop`` samples (``/root/reference/src/preprocessing.py:597-610``) and
``CoughDetectorInference.process_audio_chunk`` smooths / debounces the per-window probabilities
(``/root/reference/src/inference.py:191-241``).  Here S streams share the device:

* samples live in device ring buffers (K6 ``cough_ring_write``); the host keeps only two absolute sample
  counters per stream (written, next window start);
* each ``push`` assembles every window that just completed, over all streams, with ONE
  ``cough_window_gather`` and ONE ``cough_pipeline_forward`` (featurise with fused normalisation and stem,
  residual blocks, head);
* the only host<->device traffic per tick is the chunk upload and one (n_windows,) probability download;
* smoothing (deque mean), threshold, debounce and the "drop the rest of this chunk's windows after a
  detection" rule are the reference's, per stream, on an injectable clock;
* the steady state -- every stream pushed together with a fixed chunk length, at most one window per stream per
  tick -- is launch-bound (five small kernels and three copies), so it is captured once into two HIP graphs
  (chunk upload + ring write; the same + gather + pipeline + probability download) and replayed per tick: the
  write positions / window starts travel through one pinned int64 buffer, nothing else changes between ticks.
"""
from __future__ import annotations

from collections import deque
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .pipeline import CoughPipeline
from .preprocessing import AudioPreprocessor

SHIPPED_FLAGS = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)


class MultiStreamDetector:
    SPIN_QUERIES = 200        # bounded poll of the tick's completion event before falling back to a blocking wait

    def __init__(self, model, n_streams: int, sample_rate: int = 16000, window_duration: float = 1.0,
                 hop_duration: float = 0.25, confidence_threshold: float = 0.5, smoothing_window: int = 3,
                 debounce_seconds: float = 0.5, clock: Optional[Callable[[], float]] = None,
                 max_chunk: int = 16000, preprocessor: Optional[AudioPreprocessor] = None, use_graphs: bool = True,
                 prob_history: int = 1024):
        if not torch.cuda.is_available():
            raise RuntimeError("cough_detector_amd needs an AMD GPU (gfx950); there is no CPU fallback")
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.model = model.to(self.dev).eval()
        self.pre = preprocessor or AudioPreprocessor(sample_rate=sample_rate, segment_duration=window_duration,
                                                     device="cuda", **SHIPPED_FLAGS)
        self.pipe = CoughPipeline(self.pre, self.model)   # windows -> probabilities in one C-ABI call
        self.n_streams = n_streams
        self.window = int(sample_rate * window_duration)
        self.hop = int(sample_rate * hop_duration)
        if self.hop < 1:
            raise ValueError(f"MultiStreamDetector: hop_duration={hop_duration} gives a hop of {self.hop} samples; it must be "
                             "at least one sample")
        self.max_chunk = max_chunk
        self.ring_len = self.window + max_chunk + self.hop
        self.rings = torch.zeros((n_streams, self.ring_len), dtype=torch.float32, device=self.dev)
        self.written = np.zeros(n_streams, dtype=np.int64)       # absolute samples written per stream
        self.next_start = np.zeros(n_streams, dtype=np.int64)    # absolute start of the next window per stream
        self.threshold = confidence_threshold
        self.debounce = debounce_seconds
        self.history = [deque(maxlen=smoothing_window) for _ in range(n_streams)]
        self.last_detection = np.zeros(n_streams, dtype=np.float64)
        self.clock = clock or (lambda: __import__("time").time())
        # the most recent per-window probabilities of every stream (diagnostics) and the number of windows scored.  Plain lists,
        # trimmed to `prob_history` entries once per `prob_history` appended windows (a Python-level append hook per window costs
        # more than the GPU work of a 64-stream tick): never more than 2 x prob_history entries per stream
        if prob_history < 1:
            raise ValueError(f"MultiStreamDetector: prob_history={prob_history} must be positive")
        self.prob_history = int(prob_history)
        self.window_probs: List[List[float]] = [[] for _ in range(n_streams)]
        self.windows_seen = 0
        self._since_trim = 0
        self._lib = _lib.load()
        self.use_graphs = use_graphs
        self._g = None            # captured steady state: dict(length, buffers, graphs)

    def reset(self):
        self.written[:] = 0
        self.next_start[:] = 0
        self.last_detection[:] = 0
        for h in self.history:
            h.clear()

    # ------------------------------------------------------------------ captured steady state
    def _capture(self, length: int):
        """Two HIP graphs for "all streams, chunk of `length` samples": write only / write + one window per stream."""
        n, dev = self.n_streams, self.dev
        g = dict(length=length,
                 h_chunks=torch.empty((n, length), dtype=torch.float32).pin_memory(),
                 h_meta=torch.zeros(2 * n, dtype=torch.int64).pin_memory(),      # [write positions | window starts]
                 h_probs=torch.empty((n, 2), dtype=torch.float32).pin_memory(),
                 d_chunks=torch.empty((n, length), dtype=torch.float32, device=dev),
                 d_meta=torch.zeros(2 * n, dtype=torch.int64, device=dev),
                 d_ids=torch.arange(n, dtype=torch.int32, device=dev),
                 windows=torch.zeros((n, self.window), dtype=torch.float32, device=dev))

        def write():
            st = torch.cuda.current_stream(dev).cuda_stream
            g["d_chunks"].copy_(g["h_chunks"], non_blocking=True)
            g["d_meta"].copy_(g["h_meta"], non_blocking=True)
            _lib.check(self._lib.cough_ring_write(self.rings.data_ptr(), self.ring_len, g["d_chunks"].data_ptr(), length,
                                                  g["d_ids"].data_ptr(), g["d_meta"].data_ptr(), n, st), "cough_ring_write")

        def classify():
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(self._lib.cough_window_gather(self.rings.data_ptr(), self.ring_len, g["d_ids"].data_ptr(),
                                                     g["d_meta"][n:].data_ptr(), n, self.window,
                                                     g["windows"].data_ptr(), st), "cough_window_gather")
            _, probs, _, _ = self.pipe._run(g["windows"], True, True, False)   # no int64 cast, no strided slice:
            g["h_probs"].copy_(probs, non_blocking=True)                        # the graph holds kernels + 3 copies only

        rings_before = self.rings.clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):              # warm-up outside capture: allocators, workspaces, lazy handles
            for _ in range(2):
                write()
                classify()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g["write"], g["full"] = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g["write"]):
            write()
        with torch.cuda.graph(g["full"]):
            write()
            classify()
        self.rings.copy_(rings_before)             # the warm-up wrote h_chunks garbage at position 0
        g["done"] = torch.cuda.Event()
        g["done"].record()
        torch.cuda.synchronize(dev)
        return g

    def _push_graph(self, chunks: torch.Tensor, length: int) -> Optional[np.ndarray]:
        """Steady-state tick through the captured graphs.  Returns the (n_streams,) cough probabilities when a
        window completed on every stream, an empty array when none did, None when the tick does not fit the
        captured shape (the caller then takes the eager path)."""
        n = self.n_streams
        after = self.written + length
        ready = after - self.next_start >= self.window
        two = after - (self.next_start + self.hop) >= self.window            # a second window in the same tick
        if ready.any() != ready.all() or two.any():
            return None
        if self._g is None or self._g["length"] != length:
            try:
                self._g = self._capture(length)
            except Exception as exc:                                        # graphs unavailable: stay eager
                import warnings
                warnings.warn(f"MultiStreamDetector: HIP graph capture failed ({exc}); using eager launches")
                self.use_graphs = False
                return None
        g = self._g
        g["done"].synchronize()                    # the previous replay has consumed the pinned staging buffers
        g["h_chunks"].copy_(chunks)
        meta = g["h_meta"].numpy()
        meta[:n] = self.written
        meta[n:] = self.next_start
        self.written += length
        if not ready.all():
            g["write"].replay()                                             # no host sync on ticks without windows
            g["done"].record()
            return np.empty(0, dtype=np.float32)
        g["full"].replay()
        g["done"].record()
        self.next_start += self.hop
        # The one host sync of the tick.  The graph's GPU time is ~0.1 ms and an interrupt-driven wait costs ~50 us,
        # so the event is polled -- but only for a bounded number of queries (~0.3 ms): with one rank per GPU on a
        # shared cgroup (8 ranks on 16 cores, next to RCCL's proxy threads) an unbounded spin would burn a core
        # whenever the device is slow; past the bound the host sleeps in event.synchronize().
        for _ in range(self.SPIN_QUERIES):
            if g["done"].query():
                break
        else:
            g["done"].synchronize()
        return g["h_probs"].numpy()[:, 1].copy()

    def _decide(self, w_ids, p) -> List[Tuple[int, float, float]]:
        """Smoothing / threshold / debounce per stream (src/inference.py:219-241) for the windows of one tick."""
        detections = []
        now = self.clock()
        fired = set()
        for k, s in enumerate(w_ids):
            if s in fired:                               # inference.py:239: later windows of this chunk are dropped
                continue
            conf = float(p[k])
            self.window_probs[s].append(conf)
            h = self.history[s]
            h.append(conf)
            # np.mean of the reference (inference.py:223) adds < 8 float64 values left to right, as sum() does;
            # calling it per stream costs ~5 us, i.e. more than the GPU work of a 64-stream tick
            smoothed = sum(h) / len(h) if len(h) < 8 else float(np.mean(h))
            if smoothed >= self.threshold and now - self.last_detection[s] >= self.debounce:
                self.last_detection[s] = now
                detections.append((int(s), now, smoothed))
                fired.add(s)
        self.windows_seen += len(w_ids)
        self._since_trim += len(w_ids)
        if self._since_trim >= self.prob_history:        # a stream gained at most _since_trim entries since the last trim
            self._since_trim = 0
            for log in self.window_probs:
                if len(log) > self.prob_history:
                    del log[:len(log) - self.prob_history]
        return detections

    def push(self, chunks, stream_ids=None) -> List[Tuple[int, float, float]]:
        """chunks: (n, L) float32 (numpy or torch, host or device), one row per stream in ``stream_ids``
        (default: all streams in order).  Returns [(stream, time, smoothed_confidence), ...] detections."""
        if isinstance(chunks, np.ndarray):
            chunks = torch.from_numpy(np.ascontiguousarray(chunks, dtype=np.float32))
        if chunks.dim() == 1:
            chunks = chunks.unsqueeze(0)
        n, length = chunks.shape
        ids = np.arange(self.n_streams, dtype=np.int32) if stream_ids is None else np.asarray(stream_ids, np.int32)
        if len(ids) != n or length > self.max_chunk:
            raise ValueError("push: one row per stream id, chunk length <= max_chunk")
        if self.use_graphs and stream_ids is None and n == self.n_streams and not chunks.is_cuda and length <= self.hop:
            p = self._push_graph(chunks, length)
            if p is not None:
                return self._decide(list(range(n)), p) if len(p) else []
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        d_chunks = chunks.to(self.dev, torch.float32, non_blocking=True).contiguous()
        meta = torch.from_numpy(np.concatenate([ids.astype(np.int64), self.written[ids]])).to(self.dev, non_blocking=True)
        d_ids = meta[:n].to(torch.int32)
        _lib.check(self._lib.cough_ring_write(self.rings.data_ptr(), self.ring_len, d_chunks.data_ptr(), length,
                                              d_ids.data_ptr(), meta[n:].data_ptr(), n, stream), "cough_ring_write")
        self.written[ids] += length

        # every window that is complete now, stream-major then time order (the reference's while-loop)
        w_ids, w_starts = [], []
        for s in ids:
            while self.written[s] - self.next_start[s] >= self.window:
                w_ids.append(s)
                w_starts.append(self.next_start[s])
                self.next_start[s] += self.hop
        if not w_ids:
            return []
        nw = len(w_ids)
        wmeta = torch.from_numpy(np.concatenate([np.asarray(w_ids, np.int64), np.asarray(w_starts, np.int64)])).to(
            self.dev, non_blocking=True)
        wi = wmeta[:nw].to(torch.int32)
        windows = torch.empty((nw, self.window), dtype=torch.float32, device=self.dev)
        _lib.check(self._lib.cough_window_gather(self.rings.data_ptr(), self.ring_len, wi.data_ptr(),
                                                 wmeta[nw:].data_ptr(), nw, self.window, windows.data_ptr(), stream),
                   "cough_window_gather")
        _, probs = self.pipe.predict(windows, normalize=True)
        p = probs[:, 1].to("cpu").numpy()               # the one host sync of the tick

        return self._decide(w_ids, p)
