"""Host-side mirror of the reference featuriser interface, backed by the HIP kernel K1.

Same names, arguments and error behaviour as ``/root/reference/src/preprocessing.py``
(``AudioPreprocessor`` :13-550, ``RealtimePreprocessor`` :553-616, ``create_preprocessor``
:619-632); the arithmetic of ``extract_features`` runs in ``csrc/featurize.hip`` through the
C-ABI ``cough_featurize``.  PyTorch is used for device memory and streams only.

Differences a caller can observe:

* ``device`` names where results are RETURNED ("cpu" as the reference's inference engine
  passes, or "cuda"); compute is always on the MI355X -- there is no CPU fallback.
* ``extract_features`` additionally accepts ``(B, N)`` / ``(B, 1, N)`` batches and returns
  ``(B, F, T)`` with the reference's per-clip reduction semantics; like the reference it takes a waveform of ANY
  length N (T = 1 + N // hop_length): ``segment_samples`` runs on the tuned kernel, other lengths on the generic chain --
  through the same handle (the length is a launch parameter).
* Every flag of the reference constructor is implemented (pre-emphasis, delta-delta, PCEN, ``use_mfcc``,
  spectral contrast + centroid) for every geometry with ``n_fft`` in 16..2048: any ``sample_rate`` /
  ``hop_length`` / ``win_length`` / ``n_mels <= 256`` / ``n_mfcc`` / ``f_min`` / ``f_max`` / ``segment_duration``.  The
  shipped geometry runs on the tuned one-launch kernel, the others on a chain of small kernels
  (``csrc/featurize_generic.hip``); ``n_fft`` > 2048 raises ``ValueError``.
* ``use_spectral_contrast=True`` with 5 or more bands (the constructor default is 6) yields NaN rows exactly as
  the reference does (its first band is one bin wide, ``src/preprocessing.py:272-290``); a warning says so.
"""
from __future__ import annotations

import ctypes as C
import threading
import warnings
from typing import List, Optional, Tuple

import torch

from . import _lib, _tables


def _cuda_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("cough_detector_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False "
                           "and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _check_waveform(w, who: str) -> None:
    """torch.stft (the reference's T.Spectrogram) takes floating-point TENSORS only; an integer tensor (raw int16 PCM) would be
    featurised at 32768 times the amplitude, silently -- refuse it as the reference does."""
    if not isinstance(w, torch.Tensor):
        raise TypeError(f"{who}: expected a torch.Tensor, got {type(w).__name__}")
    if not w.dtype.is_floating_point:
        raise TypeError(f"{who}: expected a floating-point waveform, got {w.dtype} (scale integer PCM to [-1, 1] first)")


class AudioPreprocessor:
    """waveform -> stacked (mel | MFCC | delta [| delta-delta]) feature image on the GPU."""

    def __init__(
        self,
        sample_rate: int = 16000,
        n_mels: int = 64,
        n_fft: int = 512,
        hop_length: int = 160,
        win_length: int = 400,
        f_min: float = 100.0,
        f_max: float = 4000.0,
        segment_duration: float = 1.0,
        n_mfcc: int = 13,
        use_mfcc: bool = True,
        use_pcen: bool = True,
        use_pre_emphasis: bool = True,
        pre_emphasis_coef: float = 0.97,
        use_delta_delta: bool = True,
        use_spectral_contrast: bool = True,
        n_contrast_bands: int = 6,
        device: str = "cpu",
    ):
        self.sample_rate = sample_rate
        self.n_mels = n_mels
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.win_length = win_length
        self.f_min = f_min
        self.f_max = f_max
        self.segment_duration = segment_duration
        self.segment_samples = int(sample_rate * segment_duration)
        self.n_mfcc = n_mfcc
        self.use_mfcc = use_mfcc
        self.use_pcen = use_pcen
        self.use_pre_emphasis = use_pre_emphasis
        self.pre_emphasis_coef = pre_emphasis_coef
        self.use_delta_delta = use_delta_delta
        self.use_spectral_contrast = use_spectral_contrast
        self.n_contrast_bands = n_contrast_bands
        self.device = device

        if use_spectral_contrast and not 1 <= n_contrast_bands <= _lib.MAX_CONTRAST_BANDS:
            raise ValueError(f"AudioPreprocessor: n_contrast_bands={n_contrast_bands}: the MI355X path takes "
                             f"1..{_lib.MAX_CONTRAST_BANDS}")
        if use_spectral_contrast and n_contrast_bands >= 5:
            # src/preprocessing.py:272-290: band 0 is the single bin [1, 2); its "top 20 %" slice is empty, the mean of
            # an empty tensor is NaN, and the z-score at :300 spreads it over all n_contrast_bands + 1 rows.
            warnings.warn(f"use_spectral_contrast=True with n_contrast_bands={n_contrast_bands}: the reference's "
                          "spectral-contrast rows are NaN by construction for 5 or more bands (its first band is one "
                          "bin wide); this implementation reproduces them. Use use_spectral_contrast=False (the "
                          "shipped configuration, src/train.py:264-287) or n_contrast_bands <= 4.", stacklevel=2)
        # Geometry: the tuned one-kernel path serves the shipped 16 kHz / 512 / 160 / 400 / 64 mel / 13 MFCC / 1 s layout with
        # f_max <= sample_rate / 4; every other geometry (any n_fft in 16..2048) runs on the generic kernel chain
        # (csrc/featurize_generic.hip) -- the library picks.  What torch / torchaudio would refuse is refused here too.
        if not 16 <= n_fft <= 2048:
            raise ValueError(f"AudioPreprocessor: n_fft={n_fft}: the MI355X path implements n_fft = 16..2048 (512, the "
                             "reference's default, on the register FFT kernels; other powers of two on a radix-4 kernel; "
                             "anything else by direct DFT)")
        if not 1 <= win_length <= n_fft:
            raise ValueError(f"AudioPreprocessor: win_length={win_length} must lie in 1..n_fft (torch.stft)")
        if hop_length < 1:
            raise ValueError(f"AudioPreprocessor: hop_length={hop_length} must be positive")
        if not 1 <= n_mels <= 256:
            raise ValueError(f"AudioPreprocessor: n_mels={n_mels}: the MI355X path takes 1..256 mel bands")
        if use_mfcc and not 1 <= n_mfcc <= n_mels:
            raise ValueError("Cannot select more MFCC coefficients than # mel bins")          # torchaudio.transforms.MFCC
        if self.segment_samples <= n_fft // 2:
            raise ValueError(f"AudioPreprocessor: a segment of {self.segment_samples} samples is shorter than the reflect "
                             f"padding of torch.stft(center=True) (needs more than n_fft // 2 = {n_fft // 2})")
        # host tables, built the way torchaudio builds them (preprocessing.py:94-127)
        self._window = _tables.hann_window(win_length)
        self._mel_fb = _tables.mel_filterbank(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate)
        self._dct = _tables.dct_matrix(n_mfcc, n_mels)
        self._handle: Optional[C.c_void_p] = None
        self._handle_lock = threading.Lock()
        self._ws = {}                               # stream -> scratch of the spectral-contrast rows / the generic kernel chain
        self._resamplers = {}
        self._plain: Optional["AudioPreprocessor"] = None   # helper methods: the same transforms without pre-emphasis / contrast

    # ------------------------------------------------------------------ native handle
    def _native(self) -> C.c_void_p:
        """The ONE featuriser handle of this preprocessor.  None of its tables depends on the waveform length -- the reference's
        ``extract_features`` takes a waveform of ANY length (T = 1 + N // hop_length, ``src/preprocessing.py:432-489``) -- so the
        length is a launch parameter (``cough_featurize_any``): the constructor's segment runs on the tuned kernel where there is
        one, every other length on the generic kernel chain, with the same tables."""
        if self._handle is None:
            with self._handle_lock:
                if self._handle is None:
                    self._handle = self._create_handle(self.segment_samples)
        return self._handle

    def kernel_path(self) -> str:
        """Which kernels featurise ``segment_samples`` windows: "tuned" (one launch, the shipped sparse filterbank),
        "tuned_fullband" (one launch, any filterbank at the shipped STFT geometry), "tuned_geometry" (one launch, n_fft 512 with
        another hop / window / sample rate / segment of up to 128 frames) or "generic" (the kernel chain)."""
        return ("generic", "tuned", "tuned_fullband", "tuned_geometry")[_lib.load().cough_featurizer_path(self._native())]

    def _check_length(self, n_samples: int) -> None:
        if n_samples <= self.n_fft // 2:
            raise ValueError(f"a waveform of {n_samples} samples is shorter than the reflect padding of "
                             f"torch.stft(center=True) (needs more than n_fft // 2 = {self.n_fft // 2})")

    MAX_STREAM_WORKSPACES = 8

    def _workspace(self, need: int, dev: torch.device, stream: int) -> torch.Tensor:
        """Scratch for one launch on ``stream``.  One buffer per stream: two threads / streams sharing this preprocessor never
        share scratch (``include/cough_amd.h``: "each with its own workspace and stream"), and a buffer that is replaced goes back
        to torch's caching allocator, which hands it out again only in stream order."""
        ws = self._ws.get(stream)
        if ws is None or ws.numel() < need or ws.device != dev:
            if ws is None and len(self._ws) >= self.MAX_STREAM_WORKSPACES:
                self._ws.pop(next(iter(self._ws)))
            ws = self._ws[stream] = torch.empty(need, dtype=torch.uint8, device=dev)
        return ws

    def _create_handle(self, n_samples: int) -> C.c_void_p:
        lib = _lib.load()
        _cuda_device()
        cfg = _lib.FeatConfig(self.sample_rate, self.n_fft, self.hop_length, self.win_length, self.n_mels,
                              self.n_mfcc, n_samples, int(bool(self.use_pre_emphasis)),
                              float(self.pre_emphasis_coef), int(bool(self.use_delta_delta)),
                              int(bool(self.use_pcen)), int(bool(self.use_mfcc)),
                              int(bool(self.use_spectral_contrast)), int(self.n_contrast_bands))
        if self.use_spectral_contrast:
            edges = _tables.contrast_band_edges(self.n_contrast_bands, self.n_fft // 2 + 1)
            for k, e in enumerate(edges):
                cfg.contrast_edges[k] = int(e)
        h = C.c_void_p()
        _lib.check(lib.cough_featurizer_create(C.byref(h), C.byref(cfg), _lib.fptr(self._window),
                                               _lib.fptr(self._mel_fb), _lib.fptr(self._dct)),
                   "cough_featurizer_create")
        return h

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h is not None:
            try:
                _lib.load().cough_featurizer_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ reference helpers (host plumbing)
    def load_audio(self, path: str):
        """``torchaudio.load(path)`` for RIFF/WAVE files (src/preprocessing.py:155-166): (channels, samples) float32 in
        [-1, 1) and the sample rate.  Host plumbing (scipy.io.wavfile); integer PCM is scaled as torchaudio's
        ``normalize=True`` does (8-bit unsigned: (x - 128) / 128; 16 / 24 / 32-bit: / 2^(bits - 1)).  Other containers are outside
        the MI355X hot path: decode them to a float32 tensor and call process()."""
        import numpy as np
        try:
            from scipy.io import wavfile
            sr, data = wavfile.read(path)
        except Exception as e:      # noqa: BLE001  (not a WAVE file, or a codec scipy does not read)
            raise ValueError(f"load_audio: {path!r} is not a PCM / float WAVE file this build decodes ({e}); decode it to a "
                             "float32 tensor and call process()") from e
        if data.ndim == 1:
            data = data[:, None]
        if data.dtype == np.uint8:
            x = (data.astype(np.float32) - 128.0) / 128.0
        elif data.dtype == np.int16:
            x = data.astype(np.float32) / 32768.0
        elif data.dtype == np.int32:                 # 32-bit PCM, and 24-bit PCM that scipy left-justifies in 32 bits
            x = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
        elif data.dtype in (np.float32, np.float64):
            x = data.astype(np.float32)
        else:
            raise ValueError(f"load_audio: unsupported sample type {data.dtype}")
        return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)

    def resample(self, waveform: torch.Tensor, orig_sr: int) -> torch.Tensor:
        """T.Resample(orig_sr, sample_rate) on the GPU (``cough_resample``); (C, N) -> (C, ceil(N*sr/orig_sr))."""
        if orig_sr == self.sample_rate:
            return waveform
        import math
        dev = _cuda_device()
        if orig_sr not in self._resamplers:          # kernel table cached per source rate, like the reference (:144-153)
            kern, width, orig, new = _tables.sinc_resample_kernel(orig_sr, self.sample_rate)
            self._resamplers[orig_sr] = (kern.to(dev), width, orig, new)
        kern, width, orig, new = self._resamplers[orig_sr]
        x = waveform.to(device=dev, dtype=torch.float32).contiguous()
        if x.dim() != 2:
            raise ValueError(f"resample: expected (channels, samples), got {tuple(x.shape)}")
        rows, n = x.shape
        out_len = int(math.ceil(new * n / orig))
        out = torch.empty((rows, out_len), dtype=torch.float32, device=dev)
        if rows and out_len:
            _lib.check(_lib.load().cough_resample(x.data_ptr(), n, rows, n, kern.data_ptr(), orig, new, width,
                                                  out.data_ptr(), out_len, out_len,
                                                  torch.cuda.current_stream(dev).cuda_stream), "cough_resample")
        return out

    def _prepare(self, waveform: torch.Tensor, out_len: int, normalize: bool) -> torch.Tensor:
        dev = _cuda_device()
        w = waveform.to(device=dev, dtype=torch.float32)
        if w.dim() != 2 or w.shape[0] < 1 or w.shape[1] < 1:
            raise ValueError(f"expected (channels, samples), got {tuple(waveform.shape)}")
        if w.stride(1) != 1:
            w = w.contiguous()
        out = torch.empty((1, out_len), dtype=torch.float32, device=dev)
        _lib.check(_lib.load().cough_prepare_clip(w.data_ptr(), w.stride(0) if w.shape[0] > 1 else w.shape[1], w.shape[0],
                                                  w.shape[1], out.data_ptr(), out_len, 1 if normalize else 0,
                                                  torch.cuda.current_stream(dev).cuda_stream), "cough_prepare_clip")
        return out

    def to_mono(self, waveform: torch.Tensor) -> torch.Tensor:
        """(C, n) -> (1, n): mean over channels (src/preprocessing.py:185-197), on the GPU; returned where the input lives."""
        if waveform.shape[0] == 1:
            return waveform
        return self._prepare(waveform, waveform.shape[1], False).to(waveform.device)

    def normalize(self, waveform: torch.Tensor) -> torch.Tensor:
        """Peak-normalise a (1, n) waveform (silent no-op on an all-zero input, src/preprocessing.py:199-212), on the
        GPU.  ``add_audio`` and the pipeline do NOT call this: they set the fused-normalise flag of the featurise kernel."""
        if waveform.dim() != 2 or waveform.shape[0] != 1:
            max_val = waveform.abs().max()             # the reference also accepts other shapes: host plumbing
            return waveform / max_val if max_val > 0 else waveform
        return self._prepare(waveform, waveform.shape[1], True).to(waveform.device)

    def _rows_op(self, x: torch.Tensor, launch) -> torch.Tensor:
        """Run a row-wise kernel over the last axis of ``x`` on the GPU; the result comes back where ``x`` lives."""
        dev = _cuda_device()
        xin = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        out = torch.empty_like(xin)
        if xin.numel():
            launch(xin, out, torch.cuda.current_stream(dev).cuda_stream)
        return out.to(x.device)

    def apply_pre_emphasis(self, waveform: torch.Tensor) -> torch.Tensor:
        """y[n] = x[n] - coef * x[n-1], first sample kept (src/preprocessing.py:214-240); identity unless the preprocessor
        was built with ``use_pre_emphasis=True``.  ``extract_features`` applies it inside the featurise kernel."""
        if not self.use_pre_emphasis:
            return waveform
        if waveform.dim() != 2:
            raise ValueError(f"apply_pre_emphasis: expected (channels, samples), got {tuple(waveform.shape)}")
        return self._rows_op(waveform, lambda a, o, st: _lib.check(_lib.load().cough_pre_emphasis(
            a.data_ptr(), a.shape[1], o.data_ptr(), a.shape[1], a.shape[0], a.shape[1], float(self.pre_emphasis_coef), st),
            "cough_pre_emphasis"))

    def compute_deltas(self, features: torch.Tensor) -> torch.Tensor:
        """(channels, freq, time) -> same shape: replicate-padded central difference / 2 (src/preprocessing.py:342-356)."""
        if features.dim() < 1 or features.shape[-1] < 1:
            raise ValueError(f"compute_deltas: expected (..., time), got {tuple(features.shape)}")
        t = features.shape[-1]
        return self._rows_op(features, lambda a, o, st: _lib.check(_lib.load().cough_compute_deltas(
            a.data_ptr(), o.data_ptr(), a.numel() // t, t, st), "cough_compute_deltas"))

    def apply_pcen(self, mel_spec: torch.Tensor, alpha: float = 0.98, delta: float = 2.0, r: float = 0.5,
                   eps: float = 1e-6) -> torch.Tensor:
        """Per-channel energy normalisation of a mel POWER spectrogram (channels, n_mels, time)
        (src/preprocessing.py:305-340): 10-frame moving average, (mel / (eps + smooth)^alpha + delta)^r - delta^r."""
        if mel_spec.dim() != 3:
            raise ValueError(f"apply_pcen: expected (channels, n_mels, time), got {tuple(mel_spec.shape)}")
        t = mel_spec.shape[-1]
        return self._rows_op(mel_spec, lambda a, o, st: _lib.check(_lib.load().cough_pcen(
            a.data_ptr(), o.data_ptr(), a.numel() // max(t, 1), t, float(alpha), float(delta), float(r), float(eps), st),
            "cough_pcen"))

    def pad_or_trim(self, waveform: torch.Tensor, length: Optional[int] = None) -> torch.Tensor:
        if length is None:
            length = self.segment_samples
        n = waveform.shape[1]
        if n == length:
            return waveform
        if n > length:
            start = (n - length) // 2
            return waveform[:, start:start + length]
        left = (length - n) // 2
        return torch.nn.functional.pad(waveform, (left, length - n - left), mode="constant", value=0)

    def get_expected_time_frames(self) -> int:
        return (self.segment_samples // self.hop_length) + 1

    def _frames(self, n_samples: int) -> int:
        """Frames torch.stft(center=True) yields for n_samples: n // hop + 1 for an even n_fft (= get_expected_time_frames()
        for the segment), computed from one sample less for an odd n_fft (the padded length is then n + n_fft - 1)."""
        return (n_samples - (self.n_fft & 1)) // self.hop_length + 1

    def get_num_features(self) -> int:
        n = self.n_mels
        if self.use_mfcc:
            n += 2 * self.n_mfcc
            if self.use_delta_delta:
                n += self.n_mfcc
        if self.use_spectral_contrast:
            n += self.n_contrast_bands + 1
        return n

    # ------------------------------------------------------------------ the hot path
    def _out_device(self, like: torch.Tensor) -> torch.device:
        return torch.device("cpu") if str(self.device) == "cpu" else _cuda_device()

    def featurize_batch(self, waveforms: torch.Tensor, normalize: bool = False,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(B, N) float32 on the GPU -> (B, F, 1 + N // hop_length) float32 on the GPU (stream-ordered, no host sync);
        N = ``segment_samples`` is the tuned path, any other length runs on the generic kernel chain (same handle, the
        length is a launch parameter).  ``normalize=True`` fuses ``normalize()`` per clip into the kernel."""
        _check_waveform(waveforms, "featurize_batch")
        if waveforms.dim() != 2 or waveforms.shape[1] < 1:
            raise ValueError(f"featurize_batch: expected (B, N), got {tuple(waveforms.shape)}")
        n_samples = waveforms.shape[1]
        self._check_length(n_samples)
        dev = _cuda_device()
        w = waveforms.to(device=dev, dtype=torch.float32)
        if w.stride(1) != 1 or w.stride(0) % 4 != 0 or w.data_ptr() % 16 != 0:
            w = w.contiguous()
        b = w.shape[0]
        f, t = self.get_num_features(), self._frames(n_samples)
        if out is None:
            out = torch.empty((b, f, t), dtype=torch.float32, device=dev)
        elif tuple(out.shape) != (b, f, t) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise ValueError("featurize_batch: `out` must be a contiguous float32 (B, F, T) tensor on the GPU")
        if b == 0:
            return out
        stream = torch.cuda.current_stream(dev).cuda_stream
        stride = w.stride(0) if b > 1 else n_samples
        lib, h = _lib.load(), self._native()
        need = lib.cough_featurizer_workspace_bytes_for(h, n_samples, b)   # non-zero with spectral contrast / on the generic chain
        ws = self._workspace(need, dev, stream) if need else None
        _lib.check(lib.cough_featurize_any(h, w.data_ptr(), stride, n_samples, out.data_ptr(), b,
                                           _lib.FEAT_NORMALIZE if normalize else 0,
                                           ws.data_ptr() if need else None, need, stream), "cough_featurize_any")
        return out

    def spectrogram_batch(self, waveforms: torch.Tensor, power: float = 2.0, full_window: bool = False,
                          out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The STFT stage on its own: (B, segment_samples) -> (B, n_fft//2+1, T) float32 on the GPU.

        ``power=2.0`` with the featuriser's window is the reference's ``self.spectrogram`` (T.Spectrogram,
        preprocessing.py:131-136); ``power=1.0, full_window=True`` is the magnitude spectrogram
        ``T.SpectralCentroid`` (:137-141) forms with its default Hann(n_fft) window."""
        if waveforms.dim() != 2 or waveforms.shape[1] < 1:
            raise ValueError(f"spectrogram_batch: expected (B, N), got {tuple(waveforms.shape)}")
        n_samples = waveforms.shape[1]
        self._check_length(n_samples)
        if power not in (1.0, 2.0):
            raise ValueError("spectrogram_batch: power must be 1.0 or 2.0")
        dev = _cuda_device()
        w = waveforms.to(device=dev, dtype=torch.float32)
        if w.stride(1) != 1 or w.stride(0) % 4 != 0 or w.data_ptr() % 16 != 0:
            w = w.contiguous()
        b, shape = w.shape[0], (w.shape[0], self.n_fft // 2 + 1, self._frames(n_samples))
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=dev)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise ValueError("spectrogram_batch: `out` must be a contiguous float32 (B, 257, T) tensor on the GPU")
        if b == 0:
            return out
        flags = (_lib.SPEC_MAGNITUDE if power == 1.0 else 0) | (_lib.SPEC_FULL_WINDOW if full_window else 0)
        _lib.check(_lib.load().cough_spectrogram_any(self._native(), w.data_ptr(), w.stride(0) if b > 1 else n_samples,
                                                     n_samples, out.data_ptr(), b, flags,
                                                     torch.cuda.current_stream(dev).cuda_stream), "cough_spectrogram_any")
        return out

    def extract_features(self, waveform: torch.Tensor) -> torch.Tensor:
        """(1, N) -> (1, F, T) like the reference; also (B, N) / (B, 1, N) -> (B, F, T)."""
        _check_waveform(waveform, "extract_features")
        if waveform.dim() == 3:
            if waveform.shape[1] != 1:
                raise ValueError("extract_features: (B, C, N) input needs C == 1")
            waveform = waveform[:, 0]
        if waveform.dim() != 2:
            raise ValueError(f"extract_features: expected (1, N) or (B, N), got {tuple(waveform.shape)}")
        return self.featurize_batch(waveform).to(self._out_device(waveform))      # any length, as the reference (:432-489)

    def _without_pre_emphasis(self) -> "AudioPreprocessor":
        """extract_mel_spectrogram / extract_mfcc called on their own transform the waveform AS GIVEN: the reference applies
        pre-emphasis in extract_features only (src/preprocessing.py:455-459), not in these methods (:387-430)."""
        if not self.use_pre_emphasis:
            return self
        with self._handle_lock:
            if self._plain is None:
                self._plain = AudioPreprocessor(
                    sample_rate=self.sample_rate, n_mels=self.n_mels, n_fft=self.n_fft, hop_length=self.hop_length,
                    win_length=self.win_length, f_min=self.f_min, f_max=self.f_max, segment_duration=self.segment_duration,
                    n_mfcc=self.n_mfcc, use_mfcc=self.use_mfcc, use_pcen=self.use_pcen, use_pre_emphasis=False,
                    pre_emphasis_coef=self.pre_emphasis_coef, use_delta_delta=False, use_spectral_contrast=False,
                    n_contrast_bands=self.n_contrast_bands, device=self.device)
        return self._plain

    def extract_mel_spectrogram(self, waveform: torch.Tensor) -> torch.Tensor:
        """(1, N) -> (1, n_mels, T): src/preprocessing.py:387-412 (log-mel or PCEN rows of the waveform as given)."""
        return self._without_pre_emphasis().extract_features(waveform)[:, :self.n_mels]

    def extract_mfcc(self, waveform: torch.Tensor) -> torch.Tensor:
        """(1, N) -> (1, n_mfcc, T): src/preprocessing.py:414-430 (z-scored MFCCs of the waveform as given)."""
        if not self.use_mfcc:
            raise ValueError("extract_mfcc: this preprocessor was built with use_mfcc=False")
        return self._without_pre_emphasis().extract_features(waveform)[:, self.n_mels:self.n_mels + self.n_mfcc]

    def extract_spectral_contrast(self, waveform: torch.Tensor) -> torch.Tensor:
        """(1, N) -> (1, n_contrast_bands + 1, T): the rows src/preprocessing.py:242-303 appends."""
        if not self.use_spectral_contrast:
            raise ValueError("extract_spectral_contrast: this preprocessor was built with use_spectral_contrast=False")
        return self.extract_features(waveform)[:, -(self.n_contrast_bands + 1):]

    def prepare_clip(self, waveform: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """to_mono -> normalize -> pad_or_trim of ``process`` (src/preprocessing.py:505-512) in one kernel:
        (C, n) on the GPU -> (1, segment_samples) on the GPU.  The peak is that of the whole mono signal, as in the
        reference (normalisation comes before the trim)."""
        return self._prepare(waveform, self.segment_samples, normalize)

    def process(self, waveform: torch.Tensor, orig_sr: int) -> torch.Tensor:
        """resample -> mono -> normalize -> pad/trim -> features, as src/preprocessing.py:491-517: ``cough_resample``,
        ``cough_prepare_clip`` and the featurise kernel; nothing runs on the host or in torch ops."""
        clip = self.prepare_clip(self.resample(waveform, orig_sr), normalize=True)
        return self.featurize_batch(clip, normalize=False).to(self._out_device(waveform))

    def process_file(self, path: str) -> torch.Tensor:
        return self.process(*self.load_audio(path))


class RecentLog(list):
    """A list that keeps only the most recent ``maxlen`` appended values (amortised O(1) append): the per-window
    probability logs of the engines are diagnostics, and a detector that runs for months must not grow with them."""

    def __init__(self, maxlen: int = 4096):
        super().__init__()
        if maxlen < 1:
            raise ValueError(f"RecentLog: maxlen={maxlen} must be positive")
        self.maxlen = int(maxlen)

    def append(self, value) -> None:
        super().append(value)
        if len(self) >= 2 * self.maxlen:
            del self[:len(self) - self.maxlen]


class RealtimePreprocessor(AudioPreprocessor):
    """Sliding-window front end: append chunks, emit one feature image per complete window.
    All complete windows of a call are featurised by ONE kernel launch (batch = windows)."""

    def __init__(self, window_duration: float = 1.0, hop_duration: float = 0.5, **kwargs):
        super().__init__(segment_duration=window_duration, **kwargs)
        self.window_duration = window_duration
        self.hop_duration = hop_duration
        self.window_samples = int(self.sample_rate * window_duration)
        self.hop_samples = int(self.sample_rate * hop_duration)
        if self.hop_samples < 1:
            # the reference's add_audio would never advance its buffer (an endless `while`, src/preprocessing.py:597-610)
            raise ValueError(f"RealtimePreprocessor: hop_duration={hop_duration} gives a hop of {self.hop_samples} samples; "
                             "it must be at least one sample")
        self.buffer = torch.zeros(1, 0)

    def take_windows(self, audio_chunk: torch.Tensor) -> Optional[torch.Tensor]:
        """Append a chunk; return the raw complete windows as (n, window_samples) or None."""
        if audio_chunk.dim() == 1:
            audio_chunk = audio_chunk.unsqueeze(0)
        self.buffer = torch.cat([self.buffer, audio_chunk.detach().to("cpu", torch.float32)], dim=1)
        n_avail = self.buffer.shape[1]
        if n_avail < self.window_samples:
            return None
        n_win = (n_avail - self.window_samples) // self.hop_samples + 1
        windows = self.buffer[0].unfold(0, self.window_samples, self.hop_samples)[:n_win].contiguous()
        self.buffer = self.buffer[:, n_win * self.hop_samples:]
        return windows

    def add_audio(self, audio_chunk: torch.Tensor) -> List[torch.Tensor]:
        windows = self.take_windows(audio_chunk)
        if windows is None:
            return []
        feats = self.featurize_batch(windows, normalize=True).to(self._out_device(windows))
        return [feats[i:i + 1] for i in range(feats.shape[0])]

    def reset(self):
        self.buffer = torch.zeros(1, 0)


def create_preprocessor(realtime: bool = False, **kwargs) -> AudioPreprocessor:
    if realtime:
        return RealtimePreprocessor(**kwargs)
    return AudioPreprocessor(**kwargs)
