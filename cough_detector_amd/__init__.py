"""MI355X-native (gfx950) hot path of the cough detector: audio featuriser + CoughDetectorResidual
forward + sliding-window engine, behind the reference's Python call surface.  All arithmetic runs in
hand-written HIP kernels (``csrc/``) reached through the C-ABI in ``include/cough_amd.h``."""
from .preprocessing import AudioPreprocessor, RealtimePreprocessor, create_preprocessor
from .model import (CoughDetector, CoughDetectorResidual, CoughDetectorSmall, ConvBlock, ResidualBlock, create_model,
                    count_parameters)
from .inference import CoughDetectorInference, RealtimeQueueDetector
from .pipeline import CoughPipeline

__all__ = ["AudioPreprocessor", "RealtimePreprocessor", "create_preprocessor", "CoughDetectorResidual",
           "CoughDetector", "CoughDetectorSmall", "ConvBlock",
           "ResidualBlock", "create_model", "count_parameters", "CoughDetectorInference", "RealtimeQueueDetector",
           "CoughPipeline"]
