import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cough_detector_amd as cda
from cough_detector_amd import synth
sd = synth.random_state_dict(seed=3)
cfg = dict(model_type="residual", sample_rate=16000, n_mels=64, n_fft=512, hop_length=160, win_length=400, f_min=100.0,
           f_max=4000.0, segment_duration=1.0, n_mfcc=13, use_mfcc=True, use_pcen=False, use_pre_emphasis=False,
           pre_emphasis_coef=0.97, use_delta_delta=False, use_spectral_contrast=False, n_contrast_bands=6)
path = "/tmp/engine_lat.pt"
torch.save({"model_state_dict": sd, "config": cfg}, path)
eng = cda.CoughDetectorInference(path, confidence_threshold=0.7, smoothing_window=3, debounce_seconds=0.5, verbose=False)
stream = synth.make_stream(9, 60.0)
for rep in range(2):
    eng.reset(); lat = []
    for i in range(0, len(stream) - 1600 + 1, 1600):
        t0 = time.perf_counter(); eng.process_audio_chunk(stream[i:i + 1600]); lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat)
    print(f"rep {rep}: chunks {len(lat)}  p50 {np.percentile(lat, 50):.3f} ms  p90 {np.percentile(lat, 90):.3f}  p99 {np.percentile(lat, 99):.3f}  max {lat.max():.3f}; windows {len(eng.window_probs)}")
