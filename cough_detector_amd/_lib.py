"""ctypes binding of ``libcough_amd.so`` (the C-ABI declared in ``include/cough_amd.h``).

There is no CPU fallback: if the shared object is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
# COUGH_AMD_LIB: alternative build of the same ABI (same-box A/B timing of kernel variants)
LIB_PATH = os.environ.get("COUGH_AMD_LIB") or os.path.join(HERE, "libcough_amd.so")

OK, EINVAL, EUNSUPPORTED, EHIP, EWORKSPACE = 0, 1, 2, 3, 4
FEAT_NORMALIZE = 1
PATH_GENERIC, PATH_TUNED, PATH_TUNED_FULLBAND, PATH_TUNED_GEOMETRY = 0, 1, 2, 3
SPEC_MAGNITUDE, SPEC_FULL_WINDOW = 1, 2
DTYPE_FP32, DTYPE_BF16, DTYPE_BF16X3 = 0, 1, 3
DTYPES = {"fp32": DTYPE_FP32, "bf16_approx": DTYPE_BF16, "bf16x3": DTYPE_BF16X3}
APPROX_NOTE = ("compute_dtype='bf16' selects the APPROXIMATE single-bf16 mode (bf16 operands and activations): at a trained "
               "head's scale its logits are 0.05-0.3 away from the f32 reference, far outside the 1e-3 parity tolerance. "
               "Pass 'bf16_approx' to say that is intended; the parity-grade modes are 'bf16x3' (residual net) and 'fp32'.")


def normalize_dtype(compute_dtype: str, allowed) -> str:
    """'bf16' is kept as an alias of 'bf16_approx' that warns: nobody gets the approximate mode without being told."""
    if compute_dtype == "bf16" and "bf16_approx" in allowed:
        import warnings
        warnings.warn(APPROX_NOTE, UserWarning, stacklevel=4)
        return "bf16_approx"
    if compute_dtype not in allowed:
        names = ", ".join(repr(a) for a in allowed if not a.startswith("_"))
        raise ValueError(f"compute_dtype must be one of {names}, got {compute_dtype!r}")
    return compute_dtype

# every symbol include/cough_amd.h declares (tests check the library exports all of them)
SYMBOLS = (
    "cough_amd_abi_version", "cough_amd_arch", "cough_amd_last_error",
    "cough_featurizer_create", "cough_featurizer_destroy", "cough_featurizer_num_features",
    "cough_featurizer_num_frames", "cough_featurizer_path", "cough_featurize", "cough_featurizer_workspace_bytes", "cough_featurize_ws",
    "cough_spectrogram", "cough_featurizer_num_frames_for", "cough_featurizer_workspace_bytes_for", "cough_featurize_any",
    "cough_spectrogram_any",
    "cough_resnet_create", "cough_resnet_create_ex", "cough_resnet_destroy", "cough_resnet_workspace_bytes",
    "cough_resblock_create", "cough_resblock_destroy", "cough_resblock_workspace_bytes", "cough_resblock_out_shape",
    "cough_resblock_forward",
    "cough_resnet_forward", "cough_resnet_read_activation",
    "cough_cnn_create", "cough_cnn_destroy", "cough_cnn_workspace_bytes", "cough_cnn_forward", "cough_cnn_conv_output",
    "cough_pipeline_workspace_bytes", "cough_pipeline_forward",
    "cough_mask_axes", "cough_prepare_clip", "cough_resample", "cough_ring_write", "cough_window_gather",
    "cough_synth_clips", "cough_pre_emphasis", "cough_compute_deltas", "cough_pcen",
)


MAX_CONTRAST_BANDS = 16


class FeatConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("n_fft", C.c_int), ("hop_length", C.c_int), ("win_length", C.c_int),
                ("n_mels", C.c_int), ("n_mfcc", C.c_int), ("segment_samples", C.c_int),
                ("use_pre_emphasis", C.c_int), ("pre_emphasis_coef", C.c_float), ("use_delta_delta", C.c_int),
                ("use_pcen", C.c_int), ("use_mfcc", C.c_int), ("use_spectral_contrast", C.c_int),
                ("n_contrast_bands", C.c_int), ("contrast_edges", C.c_int * (MAX_CONTRAST_BANDS + 2))]


_FP = C.POINTER(C.c_float)


class ConvBN(C.Structure):
    _fields_ = [("w", _FP), ("b", _FP), ("bn_w", _FP), ("bn_b", _FP), ("bn_mean", _FP), ("bn_var", _FP)]


class ResBlockWeights(C.Structure):
    _fields_ = [("conv1", ConvBN), ("conv2", ConvBN), ("skip", ConvBN)]


class ResNetWeights(C.Structure):
    _fields_ = [("stem", ConvBN), ("block", ResBlockWeights * 2), ("fc_w", _FP), ("fc_b", _FP), ("bn_eps", C.c_float)]


class CnnBlock(C.Structure):
    _fields_ = [("cin", C.c_int), ("cout", C.c_int), ("ksize", C.c_int), ("dw_w", _FP), ("dw_b", _FP),
                ("conv", ConvBN), ("pool", C.c_int)]


class CnnWeights(C.Structure):
    _fields_ = [("n_blocks", C.c_int), ("blocks", C.POINTER(CnnBlock)), ("hidden", C.c_int), ("fc1_w", _FP),
                ("fc1_b", _FP), ("fc2_w", _FP), ("fc2_b", _FP), ("bn_eps", C.c_float)]


_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load (once) and type the library; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -m cough_detector_amd.build` "
                "(needs hipcc / ROCm, target gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        vp, ll, i = C.c_void_p, C.c_longlong, C.c_int
        lib.cough_amd_abi_version.restype = i
        lib.cough_amd_arch.restype = C.c_char_p
        lib.cough_amd_last_error.restype = C.c_char_p
        lib.cough_featurizer_create.argtypes = [C.POINTER(vp), C.POINTER(FeatConfig), _FP, _FP, _FP]
        lib.cough_featurizer_destroy.argtypes = [vp]
        lib.cough_featurizer_destroy.restype = None
        lib.cough_featurizer_num_features.argtypes = [vp]
        lib.cough_featurizer_num_frames.argtypes = [vp]
        lib.cough_featurizer_path.argtypes = [vp]
        lib.cough_featurize.argtypes = [vp, vp, ll, vp, i, i, vp]
        lib.cough_spectrogram.argtypes = [vp, vp, ll, vp, i, i, vp]
        lib.cough_featurizer_workspace_bytes.argtypes = [vp, i]
        lib.cough_featurizer_workspace_bytes.restype = C.c_size_t
        lib.cough_featurize_ws.argtypes = [vp, vp, ll, vp, i, i, vp, C.c_size_t, vp]
        lib.cough_featurizer_num_frames_for.argtypes = [vp, i]
        lib.cough_featurizer_workspace_bytes_for.argtypes = [vp, i, i]
        lib.cough_featurizer_workspace_bytes_for.restype = C.c_size_t
        lib.cough_featurize_any.argtypes = [vp, vp, ll, i, vp, i, i, vp, C.c_size_t, vp]
        lib.cough_spectrogram_any.argtypes = [vp, vp, ll, i, vp, i, i, vp]
        lib.cough_resnet_create.argtypes = [C.POINTER(vp), C.POINTER(ResNetWeights), i]
        lib.cough_resnet_create_ex.argtypes = [C.POINTER(vp), i, C.POINTER(i), C.POINTER(ConvBN), C.POINTER(ResBlockWeights),
                                               _FP, _FP, C.c_float, i]
        lib.cough_resblock_create.argtypes = [C.POINTER(vp), i, i, i, C.POINTER(ConvBN), C.POINTER(ConvBN), C.POINTER(ConvBN),
                                              C.c_float]
        lib.cough_resblock_destroy.argtypes = [vp]
        lib.cough_resblock_destroy.restype = None
        lib.cough_resblock_workspace_bytes.argtypes = [vp, i, i, i]
        lib.cough_resblock_workspace_bytes.restype = C.c_size_t
        lib.cough_resblock_out_shape.argtypes = [vp, i, i, C.POINTER(i), C.POINTER(i)]
        lib.cough_resblock_forward.argtypes = [vp, vp, i, i, i, vp, vp, C.c_size_t, vp]
        lib.cough_resnet_destroy.argtypes = [vp]
        lib.cough_resnet_destroy.restype = None
        lib.cough_resnet_workspace_bytes.argtypes = [vp, i, i, i]
        lib.cough_resnet_workspace_bytes.restype = C.c_size_t
        lib.cough_resnet_forward.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, C.c_size_t, vp]
        lib.cough_resnet_read_activation.argtypes = [vp, vp, i, i, i, i, vp, vp]
        lib.cough_cnn_create.argtypes = [C.POINTER(vp), C.POINTER(CnnWeights), i]
        lib.cough_cnn_destroy.argtypes = [vp]
        lib.cough_cnn_destroy.restype = None
        lib.cough_cnn_workspace_bytes.argtypes = [vp, i, i, i]
        lib.cough_cnn_workspace_bytes.restype = C.c_size_t
        lib.cough_cnn_forward.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, C.c_size_t, vp]
        lib.cough_cnn_conv_output.argtypes = [vp, vp, i, i, i, vp, vp, C.c_size_t, vp]
        lib.cough_pipeline_workspace_bytes.argtypes = [vp, vp, i]
        lib.cough_pipeline_workspace_bytes.restype = C.c_size_t
        lib.cough_pipeline_forward.argtypes = [vp, vp, vp, ll, i, i, vp, vp, vp, vp, vp, C.c_size_t, vp, vp, vp]
        lib.cough_mask_axes.argtypes = [vp, vp, ll, i, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), vp]
        lib.cough_prepare_clip.argtypes = [vp, ll, i, i, vp, i, i, vp]
        lib.cough_resample.argtypes = [vp, ll, i, i, vp, i, i, i, vp, ll, i, vp]
        lib.cough_ring_write.argtypes = [vp, i, vp, i, vp, vp, i, vp]
        lib.cough_window_gather.argtypes = [vp, i, vp, vp, i, i, vp, vp]
        lib.cough_synth_clips.argtypes = [vp, ll, i, ll, ll, vp]
        lib.cough_pre_emphasis.argtypes = [vp, ll, vp, ll, i, i, C.c_float, vp]
        lib.cough_compute_deltas.argtypes = [vp, vp, ll, i, vp]
        lib.cough_pcen.argtypes = [vp, vp, ll, i, C.c_float, C.c_float, C.c_float, C.c_float, vp]
        if lib.cough_amd_abi_version() != 5:
            raise RuntimeError("libcough_amd.so ABI version mismatch; rebuild it")
        _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    """Map a C status to the reference's exception convention
    (ValueError for argument/config errors, as src/model.py:313-314; RuntimeError otherwise)."""
    if status == OK:
        return
    msg = load().cough_amd_last_error().decode("utf-8", "replace")
    if status in (EINVAL, EUNSUPPORTED):
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg} (status {status})")


def fptr(t):
    """Host float32 pointer of a contiguous CPU torch tensor / numpy array (caller keeps it alive)."""
    import numpy as np
    if hasattr(t, "data_ptr"):
        return C.cast(t.data_ptr(), _FP)
    assert isinstance(t, np.ndarray) and t.dtype == np.float32 and t.flags["C_CONTIGUOUS"]
    return t.ctypes.data_as(_FP)
