"""waveform -> logits in one call (``cough_pipeline_forward``).

The reference does this per window in two steps -- ``preprocessor.add_audio`` /
``extract_features`` then ``model(x)`` (``/root/reference/src/inference.py:214-217``).  With a bf16
classifier the stem convolution runs inside the featurise kernel, so the 90x101 feature image never leaves
the CU unless ``return_features=True``.  Results equal ``model(preprocessor.featurize_batch(w)[:, None])``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from .model import CoughDetectorResidual
from .preprocessing import AudioPreprocessor, _cuda_device


class CoughPipeline:
    def __init__(self, preprocessor: AudioPreprocessor, model: torch.nn.Module):
        self.pre, self.model = preprocessor, model
        self._ws = {}   # stream -> workspace: threads / streams sharing this pipeline never share scratch

    def _run(self, waveforms: torch.Tensor, normalize: bool, want_probs: bool, return_features: bool,
             events: Optional[Tuple[torch.cuda.Event, torch.cuda.Event]] = None):
        if self.model.training:
            raise RuntimeError("CoughPipeline is inference-only: call model.eval()")
        n_samples = self.pre.segment_samples
        if waveforms.dim() != 2 or waveforms.shape[1] != n_samples:
            raise ValueError(f"expected (B, {n_samples}) waveforms, got {tuple(waveforms.shape)}")
        dev = _cuda_device()
        w = waveforms.to(device=dev, dtype=torch.float32)
        if w.stride(1) != 1 or w.stride(0) % 4 != 0 or w.data_ptr() % 16 != 0:
            w = w.contiguous()
        b = w.shape[0]
        logits = torch.empty((b, 2), dtype=torch.float32, device=dev)
        probs = torch.empty((b, 2), dtype=torch.float32, device=dev) if want_probs else None
        preds = torch.empty((b,), dtype=torch.int32, device=dev) if want_probs else None
        feats = (torch.empty((b, self.pre.get_num_features(), self.pre._frames(self.pre.segment_samples)),
                             dtype=torch.float32, device=dev) if return_features else None)
        if b and not isinstance(self.model, CoughDetectorResidual):
            # conv-stack classifiers (CoughDetector / CoughDetectorSmall): featurise, then cough_cnn_forward
            if feats is None:
                feats = torch.empty((b, self.pre.get_num_features(), self.pre._frames(self.pre.segment_samples)),
                                    dtype=torch.float32, device=dev)
            if events:
                events[0].record()
            self.pre.featurize_batch(w, normalize=normalize, out=feats)
            if events:
                events[1].record()
            if want_probs:
                logits, probs, preds = self.model._run(feats[:, None], True)
            else:
                logits = self.model._run(feats[:, None], False)
            return logits, probs, preds, (feats if return_features else None)
        if b:
            lib, fh, mh = _lib.load(), self.pre._native(), self.model._native()
            need = lib.cough_pipeline_workspace_bytes(fh, mh, b)
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = self._ws.get(stream)
            if ws is None or ws.numel() < need or ws.device != dev:
                if ws is None and len(self._ws) >= 8:
                    self._ws.pop(next(iter(self._ws)))
                ws = self._ws[stream] = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.cough_pipeline_forward(
                fh, mh, w.data_ptr(), w.stride(0) if b > 1 else n_samples, b, _lib.FEAT_NORMALIZE if normalize else 0,
                feats.data_ptr() if return_features else None, logits.data_ptr(),
                probs.data_ptr() if want_probs else None, preds.data_ptr() if want_probs else None,
                ws.data_ptr(), ws.numel(), stream,
                events[0].cuda_event if events else None, events[1].cuda_event if events else None),
                "cough_pipeline_forward")
        return logits, probs, preds, feats

    def __call__(self, waveforms: torch.Tensor, normalize: bool = True, return_features: bool = False, events=None):
        """``events``: optional pair of already-recorded ``torch.cuda.Event(enable_timing=True)`` that the library
        re-records around the featurise launch (profiling hook)."""
        logits, _, _, feats = self._run(waveforms, normalize, False, return_features, events)
        return (logits, feats) if return_features else logits

    def predict(self, waveforms: torch.Tensor, normalize: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        _, probs, preds, _ = self._run(waveforms, normalize, True, False)
        return preds.to(torch.int64), probs
