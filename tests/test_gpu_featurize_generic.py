"""Generic-geometry featuriser (csrc/featurize_generic.hip) vs the CPU oracle: every AudioPreprocessor constructor geometry
at n_fft = 512 that the tuned kernel does not cover (VERDICT r03 missing #2: /root/reference/src/preprocessing.py:32-51,
:94-127, RealtimePreprocessor(window_duration=...) :559-580, the engine's re-construction from a checkpoint's config
/root/reference/src/inference.py:89-108).  Tolerances are the north-star ones (tests/parity.py)."""
import numpy as np
import pytest
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from oracle import engine as oengine, featurizer as ofeat
from parity import FEAT_TOL, SHIPPED
from test_oracle_featurizer import GEOMETRIES, geometry_clip

pytestmark = pytest.mark.gpu

BASE = dict(sample_rate=16000, n_mels=64, hop_length=160, win_length=400, f_min=100.0, f_max=4000.0, n_mfcc=13)


def _split(geom):
    g = dict(BASE)
    g.update({k: v for k, v in geom.items() if k != "segment_duration"})
    return g, geom.get("segment_duration", 1.0)


def errors(got, ref, n_mels):
    """(abs error of the [0, 1] mel rows, |a-b| / max(|b|, 1) of the z-scored rows, strict SURVEY 8d form / max(|b|, 1e-3))"""
    got, ref = got.detach().cpu().float(), ref.detach().cpu().float()
    mel = (got[..., :n_mels, :] - ref[..., :n_mels, :]).abs().max().item()
    d = (got[..., n_mels:, :] - ref[..., n_mels:, :]).abs()
    if d.numel() == 0:
        return mel, 0.0
    return mel, (d / ref[..., n_mels:, :].abs().clamp(min=1.0)).max().item()


@pytest.mark.parametrize("name", sorted(GEOMETRIES))
def test_geometry_against_oracle(name):
    g, seconds = _split(GEOMETRIES[name])
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=seconds, **g, **SHIPPED)
    n = pre.segment_samples
    w = torch.from_numpy(np.stack([geometry_clip(s, n) for s in range(10)]))
    w[7] = 0.0                                                              # digital silence: amin clamp, no NaN
    f = pre.extract_features(w.cuda())
    nm, T = g["n_mels"], 1 + (n - g.get("n_fft", 512) % 2) // g["hop_length"]
    assert f.shape == (10, nm + 2 * g["n_mfcc"], T) == (10, pre.get_num_features(), pre._frames(n))
    assert pre.get_expected_time_frames() == 1 + n // g["hop_length"]      # the reference's helper (:532-534), n_fft-blind
    ref = ofeat.extract_features_batch(w, **ofeat.geometry_kwargs(**g))
    mel, rel = errors(f, ref, nm)
    print(f"{name}: {tuple(f.shape)} mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
    assert torch.isfinite(f).all() and mel < FEAT_TOL and rel < FEAT_TOL
    assert torch.all(f[7, :nm] == 0)
    # fused normalize() == the reference's normalize-then-extract, per clip
    raw = w * 0.31
    got = pre.featurize_batch(raw.cuda(), normalize=True)
    mel, rel = errors(got, ofeat.extract_features_batch(raw, normalize_first=True, **ofeat.geometry_kwargs(**g)), nm)
    assert mel < FEAT_TOL and rel < FEAT_TOL
    # batch invariance: the three reductions are per clip
    assert torch.equal(pre.featurize_batch(raw[3:4].cuda(), normalize=True)[0], got[3])


@pytest.mark.parametrize("flags", [dict(use_pre_emphasis=True), dict(use_delta_delta=True), dict(use_pcen=True),
                                   dict(use_mfcc=False), dict(use_pre_emphasis=True, use_delta_delta=True, use_pcen=True),
                                   dict(use_spectral_contrast=True, n_contrast_bands=3),
                                   dict(use_spectral_contrast=True, n_contrast_bands=4, use_pre_emphasis=True, use_mfcc=False)])
def test_every_constructor_flag_on_a_generic_geometry(flags):
    """2 s windows, 80 mel bands up to 8 kHz, 20 MFCCs -- with each flag of the constructor (:43-49)."""
    g = dict(BASE, n_mels=80, n_mfcc=20, f_max=8000.0)
    kw = {"use_mfcc": True, **SHIPPED, **flags}
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=2.0, **g, **kw)
    w = torch.from_numpy(np.stack([geometry_clip(s, 32000) for s in (0, 2, 3, 4, 5)]))
    ok = {k: v for k, v in kw.items()}
    f = pre.featurize_batch(w.cuda(), normalize=True)
    ref = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**g), **ok)
    assert f.shape == ref.shape == (5, pre.get_num_features(), 201)
    nbase = 80 + ((40 + (20 if kw["use_delta_delta"] else 0)) if kw["use_mfcc"] else 0)
    mel, rel = errors(f[:, :nbase], ref[:, :nbase], 80)
    print(f"{flags}: mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL
    if kw["use_spectral_contrast"]:                                  # unit-variance rows: absolute tolerance
        cerr = (f[:, nbase:].cpu() - ref[:, nbase:]).abs().max().item()
        print(f"  contrast + centroid rows: abs {cerr:.2e}")
        assert cerr < FEAT_TOL


def test_reference_default_constructor_at_another_window_length_reproduces_the_nan_rule():
    """AudioPreprocessor(segment_duration=2.0) with every other default (PCEN, pre-emphasis, delta-delta, 6 contrast bands):
    110 rows; the contrast rows are NaN by construction (>= 5 bands), the rest matches the oracle."""
    with pytest.warns(UserWarning, match="NaN by construction"):
        pre = cda.AudioPreprocessor(device="cuda", segment_duration=2.0)
    w = torch.from_numpy(np.stack([geometry_clip(s, 32000) for s in (0, 4)]))
    f = pre.extract_features(w.cuda()).cpu()
    ref = ofeat.extract_features_batch(w, use_pre_emphasis=True, use_delta_delta=True, use_pcen=True,
                                       use_spectral_contrast=True, n_contrast_bands=6)
    assert f.shape == ref.shape == (2, 110, 201)
    assert torch.isnan(f[:, 103:]).all() and torch.isnan(ref[:, 103:]).all()
    mel, rel = errors(f[:, :103], ref[:, :103], 64)
    assert mel < FEAT_TOL and rel < FEAT_TOL


@pytest.mark.parametrize("power,full_window", [(2.0, False), (1.0, True)])
def test_stft_stage_on_a_generic_geometry(power, full_window):
    """cough_spectrogram for another hop / window / length: T.Spectrogram(n_fft, win_length, hop_length) (:131-136)."""
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=0.75, hop_length=100, win_length=320, **SHIPPED)
    w = torch.from_numpy(np.stack([geometry_clip(s, 12000) for s in range(5)]))
    spec = pre.spectrogram_batch(w.cuda(), power=power, full_window=full_window).cpu()
    ref = ofeat.stft_power(w, hop=100, win=512 if full_window else 320, power=power)
    assert spec.shape == ref.shape == (5, 257, 121)
    scale = ref.amax(dim=1, keepdim=True).clamp(min=1e-20)                   # per-frame full scale
    assert ((spec - ref).abs() / scale).max().item() < 5e-6


def test_realtime_two_second_windows_through_the_engine(tmp_path):
    """RealtimePreprocessor(window_duration=2.0) rebuilt by the engine from a checkpoint's config (inference.py:89-108):
    window probabilities and detections vs the CPU engine oracle on the same 2 s windows."""
    from parity import realistic_state_dict
    sd = realistic_state_dict(11)
    cfg = dict(model_type="residual", sample_rate=16000, n_mels=64, n_fft=512, hop_length=160, win_length=400, f_min=100.0,
               f_max=4000.0, segment_duration=2.0, n_mfcc=13, use_mfcc=True, use_pcen=False, use_pre_emphasis=False,
               pre_emphasis_coef=0.97, use_delta_delta=False, use_spectral_contrast=False, n_contrast_bands=6)
    path = str(tmp_path / "two_seconds.pt")
    torch.save({"model_state_dict": sd, "config": cfg}, path)
    now = {"t": 0.0}
    eng = cda.CoughDetectorInference(path, confidence_threshold=0.5, smoothing_window=3, debounce_seconds=0.5, verbose=False,
                                     clock=lambda: now["t"])
    assert eng.preprocessor.window_samples == 32000 and eng.preprocessor.get_expected_time_frames() == 201
    ref = oengine.EngineOracle(sd, 0.5, 3, 0.5, clock=lambda: now["t"])
    ref.windower = ofeat.RealtimeWindowerOracle(window_duration=2.0, hop_duration=0.25)
    stream = synth.make_stream(9, 6.0)
    hits, ref_hits = [], []
    for i in range(0, len(stream), 1600):
        now["t"] = (i + 1600) / 16000.0
        a, b = eng.process_audio_chunk(stream[i:i + 1600]), ref.process_audio_chunk(stream[i:i + 1600])
        hits.append(a is not None)
        ref_hits.append(b is not None)
    assert len(eng.window_probs) == len(ref.window_probs) == 17
    assert np.abs(np.array(eng.window_probs) - np.array(ref.window_probs)).max() < 1e-3
    assert hits == ref_hits


def test_geometry_errors_match_what_torch_would_refuse():
    with pytest.raises(ValueError, match="n_fft"):
        cda.AudioPreprocessor(n_fft=4096, **SHIPPED)
    with pytest.raises(ValueError, match="win_length"):
        cda.AudioPreprocessor(win_length=600, **SHIPPED)
    with pytest.raises(ValueError, match="MFCC coefficients"):
        cda.AudioPreprocessor(n_mels=20, n_mfcc=30, **SHIPPED)
    with pytest.raises(ValueError, match="reflect"):
        cda.AudioPreprocessor(segment_duration=0.01, **SHIPPED)
    assert cda.AudioPreprocessor(n_mels=20, n_mfcc=30, **{**SHIPPED, "use_mfcc": False}).get_num_features() == 20


@pytest.mark.parametrize("geom,n", [(dict(segment_duration=5.0), 80000),                       # 501 frames: no LDS cap on T
                                    (dict(hop_length=600, segment_duration=0.3), 4800),         # hop > n_fft: frames do not overlap
                                    (dict(hop_length=400, segment_duration=0.02), 320),         # N < hop: ONE frame, mostly reflect padding
                                    (dict(win_length=1, segment_duration=0.25), 4000)])         # a one-tap window
def test_extreme_geometries(geom, n):
    g, seconds = _split(geom)
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=seconds, **g, **SHIPPED)
    assert pre.segment_samples == n
    w = torch.from_numpy(np.stack([geometry_clip(s, n) for s in (0, 3, 4)]))
    f = pre.featurize_batch(w.cuda(), normalize=True)
    ref = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**g))
    assert f.shape == ref.shape == (3, 90, 1 + n // g["hop_length"])
    mel, rel = errors(f, ref, 64)
    print(f"{geom}: {tuple(f.shape)} mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
    assert mel < FEAT_TOL and rel < 2 * FEAT_TOL          # (a 13-value z-score of one frame amplifies the dB error by 1 / std)


def test_multi_stream_detector_with_two_second_windows_and_hip_graphs():
    """configs[4]'s streaming engine on a generic geometry: 2 s windows / 0.25 s hop, the captured-graph steady state
    included (the generic kernel chain is launch-only: no allocation, no synchronisation inside the capture)."""
    from cough_detector_amd.streaming import MultiStreamDetector
    from parity import realistic_state_dict
    sd = realistic_state_dict(11)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="fp32")
    model.load_state_dict(sd)
    S = 4
    now = {"t": 0.0}
    det = MultiStreamDetector(model, S, window_duration=2.0, confidence_threshold=0.5, smoothing_window=3,
                              debounce_seconds=0.5, clock=lambda: now["t"])
    assert det.pre.get_expected_time_frames() == 201
    refs = [oengine.EngineOracle(sd, 0.5, 3, 0.5, clock=lambda: now["t"]) for _ in range(S)]
    for r in refs:
        r.windower = ofeat.RealtimeWindowerOracle(window_duration=2.0, hop_duration=0.25)
    streams = np.stack([synth.make_stream(30 + s, 5.0) for s in range(S)])
    got, want = [], []
    for i in range(0, streams.shape[1], 1600):
        now["t"] = (i + 1600) / 16000.0
        got.append(sorted(d[0] for d in det.push(streams[:, i:i + 1600])))
        want.append(sorted(s for s in range(S) if refs[s].process_audio_chunk(streams[s, i:i + 1600]) is not None))
    for s in range(S):
        assert len(det.window_probs[s]) == len(refs[s].window_probs) == 13
        assert np.abs(np.array(det.window_probs[s]) - np.array(refs[s].window_probs)).max() < 1e-3
    assert got == want


def test_extract_features_takes_a_waveform_of_any_length_like_the_reference():
    """/root/reference/src/preprocessing.py:432-489 never checks the length: (1, N) -> (1, F, 1 + N // hop).  One
    preprocessor (shipped geometry: its 16000-sample windows stay on the tuned kernel) featurises other lengths through
    the same handle on the generic chain (the length is a launch parameter); results vs the oracle, and the tuned path is unaffected in between."""
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    one = torch.from_numpy(geometry_clip(3, 16000))[None]
    base = pre.extract_features(one.cuda())
    for n in (12345, 40000, 257, 16001):
        w = torch.from_numpy(np.stack([geometry_clip(s, n) for s in (0, 4)]))
        f = pre.extract_features(w.cuda())
        ref = ofeat.extract_features_batch(w)
        assert f.shape == ref.shape == (2, 90, 1 + n // 160)
        mel, rel = errors(f, ref, 64)
        print(f"N = {n}: {tuple(f.shape)} mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
        assert mel < FEAT_TOL and rel < 2 * FEAT_TOL
        spec = pre.spectrogram_batch(w.cuda())
        assert spec.shape == (2, 257, 1 + n // 160)
    assert torch.equal(pre.extract_features(one.cuda()), base)      # ONE handle serves every length (ABI v5: cough_featurize_any)
    with pytest.raises(ValueError, match="reflect padding"):
        pre.extract_features(torch.zeros(1, 200))


@pytest.mark.parametrize("n_fft,flags", [(400, dict(use_pre_emphasis=True, use_spectral_contrast=True, n_contrast_bands=3)),
                                         (301, dict(use_delta_delta=True, use_pcen=True)), (1000, dict()), (32, dict(use_mfcc=False)),
                                         (2000, dict(use_pre_emphasis=True)),       # 1001 bins: 32 bin tiles in 4 passes, 147 KB of LDS
                                         (1024, dict(use_pre_emphasis=True, use_delta_delta=True, use_pcen=True)),
                                         (1024, dict(use_spectral_contrast=True, n_contrast_bands=4)),
                                         (256, dict(use_spectral_contrast=True, n_contrast_bands=3, use_pre_emphasis=True)),
                                         (64, dict(use_mfcc=False)), (2048, dict(use_delta_delta=True))])
def test_other_n_fft_with_flags(n_fft, flags):
    """n_fft other than 512 -- powers of two on the radix-2 Stockham kernel, anything else (400 = torchaudio's own default, odd
    sizes) by direct DFT: every flag, the stand-alone STFT stage included."""
    g = dict(BASE, n_fft=n_fft, win_length=min(400, n_fft), hop_length=max(16, n_fft // 4), n_mels=min(64, n_fft // 4))
    g["n_mfcc"] = min(13, g["n_mels"])
    kw = {"use_mfcc": True, **SHIPPED, **flags}
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=0.8, **g, **kw)
    w = torch.from_numpy(np.stack([geometry_clip(s, 12800) for s in (0, 2, 3, 4)]))
    f = pre.featurize_batch(w.cuda(), normalize=True)
    ref = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**g), **kw)
    nm = g["n_mels"]
    assert f.shape == ref.shape == (4, pre.get_num_features(), 1 + (12800 - n_fft % 2) // g["hop_length"])
    nbase = nm + ((2 * g["n_mfcc"] + (g["n_mfcc"] if kw["use_delta_delta"] else 0)) if kw["use_mfcc"] else 0)
    mel, rel = errors(f[:, :nbase], ref[:, :nbase], nm)
    cerr = (f[:, nbase:].cpu() - ref[:, nbase:]).abs().max().item() if kw["use_spectral_contrast"] else 0.0
    print(f"n_fft {n_fft} {flags}: mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}, contrast abs {cerr:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL and cerr < FEAT_TOL
    spec = pre.spectrogram_batch(w.cuda()).cpu()
    sref = ofeat.stft_power(w, n_fft=n_fft, hop=g["hop_length"], win=g["win_length"])
    assert spec.shape == sref.shape == (4, n_fft // 2 + 1, f.shape[2])
    scale = sref.amax(dim=1, keepdim=True).clamp(min=1e-20)
    assert ((spec - sref).abs() / scale).max().item() < 5e-6
    with pytest.raises(ValueError, match="16..2048"):
        cda.AudioPreprocessor(n_fft=4096, **SHIPPED)
