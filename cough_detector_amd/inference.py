"""Sliding-window detection engine with the reference's call surface.

Mirrors ``/root/reference/src/inference.py``: ``CoughDetectorInference`` (:39-247: checkpoint +
config contract :119-163, ``predict`` :165-189, ``process_audio_chunk`` :191-241, ``reset``
:243-247) and the CLI of ``main`` (:454-503).  Featurisation and the classifier run on the MI355X
(``preprocessing.RealtimePreprocessor`` / ``model.CoughDetectorResidual``); every window a chunk
completes goes through ONE featurise launch and ONE classifier forward.  The smoothing / debounce
state machine stays on the host and takes an injectable clock (the reference reads
``datetime.now()``, :226,233), so runs are reproducible.  Microphone back-ends (:250-451) are
hardware I/O and not part of this path; ``RealtimeQueueDetector`` keeps the queue + consumer-thread structure that drives
the path (``feed`` = the audio callback) and ``main`` reads a synthetic or ``.npy`` stream instead.
"""
from __future__ import annotations

import argparse
import queue
import threading
import time
from collections import deque
from datetime import datetime
from typing import Callable, Optional

import numpy as np
import torch

from .model import create_model
from .pipeline import CoughPipeline
from .preprocessing import RealtimePreprocessor, RecentLog


def num_features_from_config(config: dict) -> int:
    """Image height implied by a checkpoint config (inference.py:128-143)."""
    n = config.get("n_mels", 64)
    if config.get("use_mfcc", True):
        n += 2 * config.get("n_mfcc", 13)
        if config.get("use_delta_delta", True):
            n += config.get("n_mfcc", 13)
    if config.get("use_spectral_contrast", True):
        n += config.get("n_contrast_bands", 6) + 1
    return n


class CoughDetectorInference:
    def __init__(self, model_path: str, device: str = "auto", confidence_threshold: float = 0.5,
                 smoothing_window: int = 3, debounce_seconds: float = 0.5, verbose: bool = True,
                 clock: Optional[Callable[[], float]] = None, compute_dtype: str = "bf16x3", prob_history: int = 4096):
        # compute_dtype: "bf16x3" (default: split-bf16 MFMA, logits within 1e-3 of the f32 reference for all three model
        # types, several times faster) or "fp32" (exact-f32 MFMA: CPU-reference numerics)
        self.verbose = verbose
        self.confidence_threshold = confidence_threshold
        self.debounce_seconds = debounce_seconds
        self.compute_dtype = compute_dtype
        if device == "auto":
            device = "cuda"
        if str(device).startswith("cpu") or str(device) == "mps":
            raise ValueError(f"device={device!r}: this engine runs on the MI355X only (use 'auto' or 'cuda')")
        if not torch.cuda.is_available():
            raise RuntimeError("cough_detector_amd needs an AMD GPU (gfx950); there is no CPU fallback")
        self.device = torch.device(device)
        if self.verbose:
            print(f"Using device: {self.device}")
        self._load_model(model_path)
        cfg = self.config
        self.preprocessor = RealtimePreprocessor(
            sample_rate=cfg.get("sample_rate", 16000), n_mels=cfg.get("n_mels", 64), n_fft=cfg.get("n_fft", 512),
            hop_length=cfg.get("hop_length", 160), win_length=cfg.get("win_length", 400),
            f_min=cfg.get("f_min", 100.0), f_max=cfg.get("f_max", 4000.0),
            window_duration=cfg.get("segment_duration", 1.0), hop_duration=0.25,
            n_mfcc=cfg.get("n_mfcc", 13), use_mfcc=cfg.get("use_mfcc", True), use_pcen=cfg.get("use_pcen", True),
            use_pre_emphasis=cfg.get("use_pre_emphasis", True),
            pre_emphasis_coef=cfg.get("pre_emphasis_coef", 0.97),
            use_delta_delta=cfg.get("use_delta_delta", True),
            use_spectral_contrast=cfg.get("use_spectral_contrast", True),
            n_contrast_bands=cfg.get("n_contrast_bands", 6), device="cuda")
        self._pipeline = CoughPipeline(self.preprocessor, self.model)   # windows -> probabilities in one C-ABI call
        self.prediction_history = deque(maxlen=smoothing_window)
        self.last_detection_time = 0
        self.on_cough_detected: Optional[Callable[[datetime, float], None]] = None
        self._clock = clock or (lambda: datetime.now().timestamp())
        # the most recent per-window cough probabilities (diagnostics / parity tests; bounded: `prob_history` entries)
        self.window_probs = RecentLog(prob_history)
        self.windows_seen = 0

    def _load_model(self, model_path):
        if self.verbose:
            print(f"Loading model from {model_path}")
        checkpoint = torch.load(model_path, map_location="cpu", weights_only=False)
        self.config = checkpoint.get("config", {})
        model_type = self.config.get("model_type", "small")
        self.model = create_model(model_type=model_type, n_mels=num_features_from_config(self.config),
                                  num_classes=2, in_channels=1, compute_dtype=self.compute_dtype)
        self.model.load_state_dict(checkpoint["model_state_dict"])
        self.model.to(self.device)
        self.model.eval()
        if self.verbose:
            print(f"Model loaded: {model_type}")
            metrics = checkpoint.get("metrics", {})
            if metrics:
                print(f"  Validation F1: {metrics.get('f1', float('nan')):.4f}")

    @torch.no_grad()
    def predict_batch(self, spectrograms: torch.Tensor) -> torch.Tensor:
        """(n, F, T) or (n, 1, F, T) on any device -> (n,) cough probabilities on the host (one sync)."""
        if spectrograms.dim() == 3:
            spectrograms = spectrograms.unsqueeze(1)
        _, probs = self.model.predict(spectrograms.to(self.device))
        return probs[:, 1].to("cpu")

    @torch.no_grad()
    def predict(self, spectrogram: torch.Tensor) -> tuple:
        """(1, F, T) or (B, 1, F, T) -> (is_cough, confidence) of element 0, as the reference."""
        if spectrogram.dim() == 3:
            spectrogram = spectrogram.unsqueeze(0)
        cough_prob = self.predict_batch(spectrogram)[0].item()
        return cough_prob > 0.5, cough_prob

    def process_audio_chunk(self, audio_chunk) -> Optional[tuple]:
        if isinstance(audio_chunk, np.ndarray):
            audio_chunk = torch.from_numpy(audio_chunk.astype(np.float32))
        if audio_chunk.dim() == 1:
            audio_chunk = audio_chunk.unsqueeze(0)
        if audio_chunk.shape[0] > 1:
            audio_chunk = audio_chunk.mean(dim=0, keepdim=True)
        windows = self.preprocessor.take_windows(audio_chunk)
        if windows is None:
            return None
        _, p2 = self._pipeline.predict(windows, normalize=True)                  # normalise + featurise + classify
        probs = p2[:, 1].to("cpu").tolist()                                      # the one host sync of the chunk
        for confidence in probs:
            self.window_probs.append(confidence)
            self.windows_seen += 1
            self.prediction_history.append(confidence)
            smoothed = float(np.mean(self.prediction_history))
            now = self._clock()
            if smoothed >= self.confidence_threshold and now - self.last_detection_time >= self.debounce_seconds:
                self.last_detection_time = now
                timestamp = datetime.fromtimestamp(now)
                if self.on_cough_detected:
                    self.on_cough_detected(timestamp, smoothed)
                return timestamp, smoothed      # later windows of this chunk are dropped, as inference.py:239
        return None

    def reset(self):
        self.preprocessor.reset()
        self.prediction_history.clear()
        self.last_detection_time = 0


class RealtimeQueueDetector:
    """The queue-and-consumer-thread half of the reference's ``RealtimeMicrophoneDetector``
    (``/root/reference/src/inference.py:250-430``) without its audio back-ends: whatever captures audio calls ``feed`` --
    what the reference's sounddevice / pyaudio callbacks do with ``audio_queue.put(indata.copy())`` (:296-300, :382-386)
    -- and ONE consumer thread drains the queue into ``inference.process_audio_chunk`` (:302-324), the only place the GPU
    path is entered.  ``start`` / ``stop`` / ``on_detection`` keep the reference's names and behaviour; detections are also
    collected in ``detections``.  ``clock_from_samples=True`` drives the engine's clock from the number of samples
    consumed (reproducible), otherwise the engine's own clock (wall time) is used as in the reference."""

    def __init__(self, inference_engine: CoughDetectorInference, sample_rate: int = 16000, chunk_duration: float = 0.1,
                 clock_from_samples: bool = False, verbose: bool = False):
        self.inference = inference_engine
        self.sample_rate = sample_rate
        self.chunk_size = int(sample_rate * chunk_duration)
        self.running = False
        self.audio_queue: "queue.Queue" = queue.Queue()
        self.on_detection: Optional[Callable[[datetime, float], None]] = None
        self.detections = RecentLog(4096)   # (stream time or wall time, confidence), the most recent ones
        self.errors = RecentLog(256)        # the reference prints and goes on (:323-324); kept for the caller as well
        self.verbose = verbose
        self._consumed = 0
        self._clock_from_samples = clock_from_samples

    def feed(self, chunk) -> None:
        """The audio callback: enqueue a copy, nothing else."""
        self.audio_queue.put(np.array(chunk, dtype=np.float32, copy=True))

    def _process_audio(self):
        while self.running or not self.audio_queue.empty():
            try:
                audio_chunk = self.audio_queue.get(timeout=0.05)
            except queue.Empty:
                continue
            try:
                audio_chunk = audio_chunk.flatten()
                self._consumed += len(audio_chunk)
                if self._clock_from_samples:
                    t = self._consumed / float(self.sample_rate)
                    self.inference._clock = lambda t=t: t
                result = self.inference.process_audio_chunk(audio_chunk)
                if result is not None:
                    timestamp, confidence = result
                    self.detections.append((timestamp.timestamp(), confidence))
                    if self.verbose:
                        print(f"COUGH DETECTED at {timestamp.strftime('%Y-%m-%d %H:%M:%S.%f')[:-3]}  confidence {confidence:.2%}")
                    if self.on_detection:
                        self.on_detection(timestamp, confidence)
            except Exception as e:      # noqa: BLE001 -- the consumer must survive a bad chunk, as the reference's does
                self.errors.append(e)
                if self.verbose:
                    print(f"Error processing audio: {e}")

    def start(self):
        if self.running:
            return
        self.running = True
        self.inference.reset()
        self._consumed = 0
        self.process_thread = threading.Thread(target=self._process_audio, name="cough-consumer")
        self.process_thread.start()

    def stop(self, timeout: float = 60.0):
        """Stop after the queue has been drained (the reference drops what is still queued; a file / test driver wants it all)."""
        self.running = False
        if hasattr(self, "process_thread"):
            self.process_thread.join(timeout=timeout)


def main(argv=None):
    parser = argparse.ArgumentParser(description="Real-time cough detection (MI355X path)")
    parser.add_argument("--model", type=str, required=True, help="Path to trained model checkpoint")
    parser.add_argument("--threshold", type=float, default=0.7, help="Confidence threshold for detection (0-1)")
    parser.add_argument("--smoothing", type=int, default=3, help="Number of predictions to average")
    parser.add_argument("--debounce", type=float, default=0.5, help="Minimum seconds between detections")
    parser.add_argument("--device", type=str, default="auto", help="Compute device (auto, cuda)")
    parser.add_argument("--audio-device", type=int, default=None, help="(accepted for compatibility; unused)")
    parser.add_argument("--backend", type=str, default="auto", choices=["auto", "sounddevice", "pyaudio"],
                        help="(accepted for compatibility; microphone capture is not part of this build)")
    parser.add_argument("--list-devices", action="store_true", help="List available audio devices and exit")
    parser.add_argument("--quiet", action="store_true", help="Suppress verbose output")
    parser.add_argument("--input", type=str, default=None,
                        help=".npy file with a 16 kHz mono float32 stream (default: a synthetic stream)")
    parser.add_argument("--seconds", type=float, default=10.0, help="Length of the synthetic stream")
    parser.add_argument("--compute-dtype", type=str, default="bf16x3", choices=["fp32", "bf16x3", "bf16_approx"],
                        help="classifier arithmetic (bf16x3 = split-bf16 MFMA within 1e-3 of the f32 logits, the default; "
                             "fp32 = exact-f32 MFMA)")
    args = parser.parse_args(argv)
    if args.list_devices:
        print("No audio capture back-end in this build; pass --input stream.npy or use the synthetic stream.")
        return
    engine = CoughDetectorInference(model_path=args.model, device=args.device, confidence_threshold=args.threshold,
                                    smoothing_window=args.smoothing, debounce_seconds=args.debounce,
                                    verbose=not args.quiet, compute_dtype=args.compute_dtype)
    if args.input:
        stream = np.load(args.input).astype(np.float32).reshape(-1)
    else:
        from .synth import make_stream
        stream = make_stream(0, args.seconds)
    sr = engine.config.get("sample_rate", 16000)
    chunk = int(sr * 0.1)
    t0, detections = time.time(), []
    sim = {"t": 0.0}
    engine._clock = lambda: sim["t"]        # stream time, not wall time: 0.1 s per chunk
    for i in range(0, len(stream) - chunk + 1, chunk):
        sim["t"] = (i + chunk) / sr
        hit = engine.process_audio_chunk(stream[i:i + chunk])
        if hit is not None:
            detections.append((sim["t"], hit[1]))
            if not args.quiet:
                print(f"[{sim['t']:7.2f}s] COUGH DETECTED (confidence: {hit[1]:.1%})")
    if not args.quiet:
        print(f"{len(engine.window_probs)} windows, {len(detections)} detections, {time.time() - t0:.2f} s wall")
    return {"detections": detections, "window_probs": list(engine.window_probs)}


if __name__ == "__main__":
    main()
