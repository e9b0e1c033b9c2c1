"""The one-launch featuriser with a run-time STFT geometry (featurize_kernel<..., GEO = true>, COUGH_PATH_TUNED_GEOMETRY) vs the
CPU oracle.

``AudioPreprocessor(sample_rate=..., hop_length=..., win_length=..., segment_duration=...)`` at n_fft = 512
(``/root/reference/src/preprocessing.py:32-51, 94-136``; ``RealtimePreprocessor(window_duration=...)`` :559-580) used to leave
the one-launch kernel for the generic kernel chain as soon as the hop, the window or the segment length differed from the shipped
ones.  The GEO instantiations take hop (<= 256), window (<= 512), segment length (while the n_mels x frames dB buffer fits the LDS
of two workgroups per CU: 2 s windows of 64 bands) and the filterbank at run time; contrast rows and longer segments stay on the
generic chain."""
import numpy as np
import pytest
import torch

import cough_detector_amd as cda
from oracle import featurizer as ofeat
from parity import FEAT_TOL, SHIPPED
from test_oracle_featurizer import geometry_clip

pytestmark = pytest.mark.gpu

BASE = dict(sample_rate=16000, n_mels=64, n_fft=512, hop_length=160, win_length=400, f_min=100.0, f_max=4000.0, n_mfcc=13)
GEO = {      # name -> (geometry overrides, segment seconds)
    "hop200_40mel": (dict(hop_length=200, n_mels=40), 1.0),
    "half_second": (dict(), 0.5),
    "quarter_second_fmax8k": (dict(f_max=8000.0), 0.25),
    "hop128_win512_126_frames": (dict(hop_length=128, win_length=512), 1.0),
    "sr22050_hop220_win441": (dict(sample_rate=22050, f_max=8000.0, hop_length=220, win_length=441), 1.0),
    "sr8000_two_seconds_hop256": (dict(sample_rate=8000, f_max=4000.0, hop_length=256, win_length=512), 2.0),
    "odd_hop77_win37_20mel": (dict(hop_length=77, win_length=37, n_mels=20), 0.5),
    "one_tap_window": (dict(win_length=1), 0.25),
    "hop3_107_frames": (dict(hop_length=3, win_length=64, n_mels=32), 0.02),
    "shortest_segment_257_samples": (dict(hop_length=16, win_length=128, sample_rate=25700, f_max=8000.0), 0.01),
    "mel80_mfcc20_hop200": (dict(n_mels=80, n_mfcc=20, f_max=8000.0, hop_length=200), 1.0),
    "mel100_127_frames": (dict(n_mels=100, n_mfcc=16, f_min=0.0, f_max=8000.0, hop_length=126, win_length=512), 1.0),
    "mel2_128_frames": (dict(n_mels=2, n_mfcc=2, f_max=8000.0, hop_length=125, win_length=250), 0.9925),
    "hop100_161_frames": (dict(hop_length=100), 1.0),
    "two_seconds_201_frames": (dict(), 2.0),
    "two_seconds_fmax8k_20mfcc": (dict(f_max=8000.0, n_mfcc=20), 2.0),
    "five_seconds_20mel_8mfcc_501_frames": (dict(n_mels=20, n_mfcc=8), 5.0),
    "hop31_130_frames": (dict(hop_length=31, win_length=100), 0.25),
    "hop300_spans_still_cover_the_segment": (dict(hop_length=300), 1.0),
    "hop512_win512": (dict(hop_length=512, win_length=512), 1.0),
    "odd_63_mel_bands_at_the_shipped_stft": (dict(n_mels=63), 1.0),
    "odd_13_mel_13_mfcc_hop200": (dict(n_mels=13, n_mfcc=13, hop_length=200), 1.0),
    "odd_101_mel_bands_fmax8k": (dict(n_mels=101, n_mfcc=20, f_max=8000.0, f_min=0.0), 1.0),
    "mel128_mfcc40_torchaudio_default_count": (dict(n_mels=128, n_mfcc=40, f_min=20.0, f_max=7600.0), 1.0),
    "mfcc41_at_the_shipped_stft": (dict(n_mfcc=41), 1.0),
    "mfcc33_hop200": (dict(n_mfcc=33, hop_length=200), 1.0),
}


def _errors(got, ref, n_mels):
    got, ref = got.detach().cpu().float(), ref.detach().cpu().float()
    mel = (got[..., :n_mels, :] - ref[..., :n_mels, :]).abs().max().item()
    d = (got[..., n_mels:, :] - ref[..., n_mels:, :]).abs()
    return mel, ((d / ref[..., n_mels:, :].abs().clamp(min=1.0)).max().item() if d.numel() else 0.0)


def _make(name, **flags):
    over, seconds = GEO[name]
    g = {**BASE, **over}
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=seconds, **g, **{**SHIPPED, **flags})
    return pre, g


@pytest.mark.parametrize("name", sorted(GEO))
def test_runtime_geometry_against_oracle(name):
    pre, g = _make(name)
    assert pre.kernel_path() == "tuned_geometry"
    n, nm = pre.segment_samples, g["n_mels"]
    T = 1 + n // g["hop_length"]
    assert n > 256
    w = torch.from_numpy(np.stack([geometry_clip(s, n) for s in range(12)]))
    w[7] = 0.0                                                       # digital silence: amin clamp, no NaN
    w[8] = 0.25                                                      # DC: only the reflect padding and the window shape it
    w[9, : n // 2] = 0.0                                             # half silent: the top_db floor is active
    kw = ofeat.geometry_kwargs(**g)
    for normalize in (False, True):
        got = pre.featurize_batch(w.cuda(), normalize=normalize)
        ref = ofeat.extract_features_batch(w, normalize_first=normalize, **kw)
        assert got.shape == ref.shape == (12, nm + 2 * g["n_mfcc"], T) == (12, pre.get_num_features(), pre._frames(n))
        keep = [i for i in range(12) if i != 8]                      # DC is ill-conditioned in float32 (tests/parity.py)
        mel, rel = _errors(got[keep], ref[keep], nm)
        print(f"{name} normalize={normalize}: {tuple(got.shape)} mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
        assert torch.isfinite(got).all() and mel < FEAT_TOL and rel < 2 * FEAT_TOL
        melh, relh = _errors(got[8:9], ref[8:9], nm)
        assert melh < 2e-2 and relh < 2e-2
        assert torch.all(got[7, :nm] == 0)
        # batch invariance: every reduction is per clip
        assert torch.equal(pre.featurize_batch(w[3:4].cuda(), normalize=normalize)[0], got[3])
    # another length through the same handle (the length is a launch parameter of the same kernel, or the generic chain's)
    other = torch.from_numpy(np.stack([geometry_clip(s, n + 300) for s in (0, 1)]))
    f2 = pre.extract_features(other.cuda())
    mel, rel = _errors(f2, ofeat.extract_features_batch(other, **kw), nm)
    assert f2.shape[2] == 1 + (n + 300) // g["hop_length"] and mel < FEAT_TOL and rel < 2 * FEAT_TOL
    assert torch.equal(pre.featurize_batch(w.cuda(), normalize=True), got)


@pytest.mark.parametrize("flags", [dict(use_pre_emphasis=True), dict(use_delta_delta=True), dict(use_pcen=True), dict(use_mfcc=False),
                                   dict(use_pre_emphasis=True, use_delta_delta=True, use_pcen=True),
                                   dict(use_spectral_contrast=True, n_contrast_bands=4),
                                   dict(use_spectral_contrast=True, n_contrast_bands=3, use_pre_emphasis=True, use_delta_delta=True),
                                   dict(use_spectral_contrast=True, n_contrast_bands=1, use_mfcc=False)],
                         ids=["preemph", "dd", "pcen", "no_mfcc", "preemph_dd_pcen", "contrast4", "contrast3_preemph_dd",
                              "contrast1_wide_band_no_mfcc"])
@pytest.mark.parametrize("name", ["half_second", "sr22050_hop220_win441", "hop128_win512_126_frames", "odd_hop77_win37_20mel",
                                  "two_seconds_201_frames", "odd_63_mel_bands_at_the_shipped_stft", "hop200_40mel",
                                  "mel80_mfcc20_hop200", "mel128_mfcc40_torchaudio_default_count", "odd_101_mel_bands_fmax8k"])
def test_runtime_geometry_flags(name, flags):
    pre, g = _make(name, **flags)
    kw = {**SHIPPED, **flags}
    n, nm = pre.segment_samples, g["n_mels"]
    T = 1 + n // g["hop_length"]
    # never the generic chain (PCEN: thread = band x quarter of the frames, bands 64..127 in a second round); at the shipped STFT a flag
    # set the fixed-geometry full-band kernel takes (e.g. no MFCC rows with 128 bands) stays there
    shipped_stft = all(k in ("n_mels", "n_mfcc", "f_min", "f_max") for k in GEO[name][0]) and GEO[name][1] == 1.0
    want = pre.kernel_path() if shipped_stft and pre.kernel_path() == "tuned_fullband" else "tuned_geometry"
    assert pre.kernel_path() == want
    w = torch.from_numpy(np.stack([geometry_clip(s, n) for s in range(6)]))
    got = pre.featurize_batch(w.cuda(), normalize=True)
    ref = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**g), **kw)
    assert got.shape == ref.shape == (6, pre.get_num_features(), T)
    nbase = got.shape[1] - (flags["n_contrast_bands"] + 1 if flags.get("use_spectral_contrast") else 0)
    mel, rel = _errors(got[:, :nbase], ref[:, :nbase], nm)
    cerr = (got[:, nbase:].cpu() - ref[:, nbase:]).abs().max().item() if nbase < got.shape[1] else 0.0     # unit-variance rows
    print(f"{name} {flags} [{want}]: mel abs {mel:.2e}, rest rel {rel:.2e}, contrast rows abs {cerr:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL and cerr < 2e-4
    if nbase < got.shape[1]:       # batch invariance with the contrast rows' sub-batches and scratch
        assert torch.equal(pre.featurize_batch(w[2:3].cuda(), normalize=True)[0], got[2])


def test_what_stays_on_the_generic_chain():
    def path(seconds=1.0, **kw):
        return cda.AudioPreprocessor(device="cuda", segment_duration=seconds, **{**BASE, **SHIPPED, **kw}).kernel_path()
    assert path(hop_length=513) == "generic" and path(0.99, hop_length=500) == "generic"   # samples between / behind the frames' spans
    assert path(5.0) == "generic" and path(2.0, n_mels=80, f_max=8000.0) == "generic"            # 64 x 501 / 80 x 201 dB values
    assert path(2.0, n_mfcc=21, n_mels=40) == "generic"                                          # 21 x 201 MFCC values > 16 640 B
    assert path(hop_length=200, use_spectral_contrast=True, n_contrast_bands=3) == "tuned_geometry"   # rows [0, nbase) in one launch
    assert path(hop_length=200, n_mels=129, f_max=8000.0) == "generic" and path(hop_length=200, n_mfcc=52) == "generic"   # 52 x 81 x 4 B
    assert path(n_mels=80, f_max=8000.0, hop_length=200, use_pcen=True) == "tuned_geometry"
    assert path(n_mels=40, use_pcen=True) == "tuned_geometry" and path(n_mels=63, use_pcen=True) == "tuned_geometry"
    assert path(2.2, use_pcen=True) == "generic"                                                 # PCEN: up to 208 frames
    assert path(n_fft=256, win_length=256) == "generic" and path(n_fft=1024) == "generic"
    assert path(hop_length=126, win_length=512, n_mels=128, n_mfcc=16, f_max=8000.0) == "generic"     # 64 KB of mel rows in LDS
    assert path(hop_length=126, win_length=512, n_mfcc=20, f_max=8000.0) == "tuned_geometry"


def test_runtime_geometry_engine_windows(tmp_path):
    """RealtimePreprocessor(window_duration=0.5) rebuilt by the engine from a checkpoint's config (inference.py:89-108): the
    0.5 s windows take the one-launch kernel; probabilities and detections vs the CPU engine oracle."""
    from cough_detector_amd import synth
    from oracle import engine as oengine
    from parity import realistic_state_dict
    sd = realistic_state_dict(11)
    cfg = dict(model_type="residual", sample_rate=16000, n_mels=64, n_fft=512, hop_length=160, win_length=400, f_min=100.0,
               f_max=4000.0, segment_duration=0.5, n_mfcc=13, use_mfcc=True, use_pcen=False, use_pre_emphasis=False,
               pre_emphasis_coef=0.97, use_delta_delta=False, use_spectral_contrast=False, n_contrast_bands=6)
    path = str(tmp_path / "half_second.pt")
    torch.save({"model_state_dict": sd, "config": cfg}, path)
    now = {"t": 0.0}
    eng = cda.CoughDetectorInference(path, confidence_threshold=0.5, smoothing_window=3, debounce_seconds=0.5, verbose=False,
                                     clock=lambda: now["t"])
    assert eng.preprocessor.window_samples == 8000 and eng.preprocessor.kernel_path() == "tuned_geometry"
    ref = oengine.EngineOracle(sd, 0.5, 3, 0.5, clock=lambda: now["t"])
    ref.windower = ofeat.RealtimeWindowerOracle(window_duration=0.5, hop_duration=0.25)
    stream = synth.make_stream(9, 4.0)
    hits, ref_hits = [], []
    for i in range(0, len(stream), 1600):
        now["t"] = (i + 1600) / 16000.0
        a, b = eng.process_audio_chunk(stream[i:i + 1600]), ref.process_audio_chunk(stream[i:i + 1600])
        hits.append(a is not None)
        ref_hits.append(b is not None)
    assert len(eng.window_probs) == len(ref.window_probs) > 10
    assert np.abs(np.array(eng.window_probs) - np.array(ref.window_probs)).max() < 1e-3
    assert hits == ref_hits


def test_runtime_geometry_full_size_batch_properties():
    """B = 4096 at 22.05 kHz through size-independent properties: batch invariance, mel rows in [0, 1], z-scored rows mean 0 /
    unbiased std 1 per clip, delta rows = central difference of the stored MFCC rows."""
    pre, g = _make("sr22050_hop220_win441")
    from cough_detector_amd import synth
    base = synth.device_clips(0, 4096)
    w = torch.cat([base, base[:, :6050].flip(1)], dim=1).contiguous()            # 22 050 samples per clip
    f = pre.featurize_batch(w, normalize=True)
    assert f.shape == (4096, 90, 101) and torch.isfinite(f).all()
    assert torch.equal(pre.featurize_batch(w[1000:1006], normalize=True), f[1000:1006])
    mel, mf, dl = f[:, :64], f[:, 64:77], f[:, 77:90]
    assert mel.min().item() >= 0.0 and mel.max().item() <= 1.0
    flat = mf.reshape(4096, -1)
    assert flat.mean(dim=1).abs().max().item() < 1e-4 and (flat.std(dim=1) - 1).abs().max().item() < 1e-4
    pad = torch.nn.functional.pad(mf, (1, 1), mode="replicate")
    assert torch.equal(dl, (pad[:, :, 2:] - pad[:, :, :-2]) / 2)


def test_pipeline_on_a_runtime_geometry_with_contrast_rows():
    """CoughPipeline over a run-time-geometry featuriser (0.5 s windows, 3 contrast bands + centroid: a 94 x 51 image): the stem
    is not fused there and the image size is not one of the split-bf16 instantiations, so the exact-f32 kernels classify it
    (announced by effective_dtype); logits vs featurise -> oracle classifier."""
    from oracle import resnet as ores
    from parity import realistic_state_dict
    sd = realistic_state_dict(17)
    flags = {**SHIPPED, "use_spectral_contrast": True, "n_contrast_bands": 3}
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=0.5, **BASE, **flags)
    assert pre.kernel_path() == "tuned_geometry" and pre.get_num_features() == 94
    model = cda.create_model("residual", n_mels=94, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(sd)
    model.eval()
    pipe = cda.CoughPipeline(pre, model)
    w = torch.from_numpy(np.stack([geometry_clip(s, 8000) for s in range(24)]))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        logits, feats = pipe(w.cuda(), normalize=True, return_features=True)
    assert model.effective_dtype(94, 51) == "fp32" and feats.shape == (24, 94, 51)
    ref_f = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**BASE), **flags)
    assert (feats.cpu() - ref_f).abs().max().item() < 2e-4
    ref_l = ores.forward(ref_f[:, None], sd)
    err = (logits.cpu() - ref_l).abs().max().item()
    print(f"pipeline on 94 x 51 images: logits max abs err {err:.2e}")
    assert err < 1e-3 and torch.equal(logits.cpu().argmax(1), ref_l.argmax(1))


def test_runtime_geometry_more_clips_than_one_sub_batch():
    """33 000 quarter-second clips with contrast rows: the contrast kernels run in sub-batches of at most 32 768 clips behind ONE
    launch of the one-launch kernel; every clip equals its own result in a batch of four (size-independent property)."""
    flags = {**SHIPPED, "use_spectral_contrast": True, "n_contrast_bands": 2, "use_pre_emphasis": True}
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=0.25, **BASE, **flags)
    assert pre.kernel_path() == "tuned_geometry"
    from cough_detector_amd import synth
    base = synth.device_clips(0, 8250)                                   # 8250 x 16000 -> 33 000 clips of 4000 samples
    w = base.reshape(33000, 4000)
    f = pre.featurize_batch(w, normalize=True)
    assert f.shape == (33000, 93, 26) and torch.isfinite(f).all()
    for i in (0, 32766, 32768, 32999):
        a = i - i % 4 if i + 4 <= 33000 else 32996
        small = pre.featurize_batch(w[a:a + 4].contiguous(), normalize=True)
        assert torch.equal(small, f[a:a + 4]), i
    ref = ofeat.extract_features_batch(w[32990:33000].cpu(), normalize_first=True, **ofeat.geometry_kwargs(**BASE), **flags)
    assert (f[32990:].cpu() - ref).abs().max().item() < 2e-4


def test_raw_c_abi_argument_errors_on_the_runtime_geometry_path():
    """cough_featurize_any on a run-time-geometry handle with contrast rows: a missing / short / misaligned workspace and a row
    stride below the segment length come back as error codes with a message (ValueError / RuntimeError through _lib.check),
    never as a launch."""
    from cough_detector_amd import _lib
    flags = {**SHIPPED, "use_spectral_contrast": True, "n_contrast_bands": 2}
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=0.5, **BASE, **flags)
    lib, h = _lib.load(), pre._native()
    assert lib.cough_featurizer_path(h) == _lib.PATH_TUNED_GEOMETRY
    b, n = 6, 8000
    w = torch.randn(b, n, device="cuda")
    out = torch.empty(b, pre.get_num_features(), 51, device="cuda")
    need = lib.cough_featurizer_workspace_bytes_for(h, n, b)
    assert need > 0
    ws = torch.empty(need + 512, dtype=torch.uint8, device="cuda")
    base = ws.data_ptr() + (-ws.data_ptr()) % 256
    st = torch.cuda.current_stream().cuda_stream

    def call(stride=n, wsp=base, wsb=need, flags_=_lib.FEAT_NORMALIZE):
        return lib.cough_featurize_any(h, w.data_ptr(), stride, n, out.data_ptr(), b, flags_, wsp, wsb, st)
    assert call() == _lib.OK
    good = out.clone()
    for kw in (dict(wsp=None, wsb=0), dict(wsb=need - 1), dict(wsp=base + 4)):
        rc = call(**kw)
        assert rc != _lib.OK and b"workspace" in lib.cough_amd_last_error()
        with pytest.raises((RuntimeError, ValueError), match="workspace"):
            _lib.check(rc, "cough_featurize_any")
    rc = call(stride=n - 1)
    assert rc == _lib.EINVAL and b"stride" in lib.cough_amd_last_error()
    assert call(flags_=0) == _lib.OK                                      # without normalize the peaks are not needed: still fine
    assert call() == _lib.OK and torch.equal(out, good)                   # and the handle is unharmed


def test_other_waveform_lengths_through_one_handle_take_the_one_launch_kernel_too():
    """extract_features never checks the length (/root/reference/src/preprocessing.py:432-489).  A handle built for the shipped
    1 s segment featurises other lengths with the run-time-geometry instantiation when they fit its limits (here: up to ~2.2 s of
    64 bands) and with the generic chain beyond; results vs the oracle either way, flags included, and the segment's own kernel is
    untouched in between."""
    from cough_detector_amd import _lib
    for flags in (SHIPPED, {**SHIPPED, "use_pre_emphasis": True, "use_delta_delta": True},
                  {**SHIPPED, "use_spectral_contrast": True, "n_contrast_bands": 3}, {**SHIPPED, "use_pcen": True}):
        pre = cda.AudioPreprocessor(device="cuda", **BASE, **flags)
        assert pre.kernel_path() == "tuned"
        lib, h = _lib.load(), pre._native()
        one = torch.from_numpy(geometry_clip(3, 16000))[None]
        base = pre.featurize_batch(one.cuda(), normalize=True)
        for n in (8000, 12345, 16001, 30000, 257, 48000):
            w = torch.from_numpy(np.stack([geometry_clip(s, n) for s in (0, 4, 5)]))
            need = lib.cough_featurizer_workspace_bytes_for(h, n, 3)
            one_launch = n // 160 + 1 <= (208 if flags.get("use_pcen") else 222)
            if not flags.get("use_spectral_contrast"):
                assert (need == 0) == one_launch, (n, need)              # the generic chain always needs scratch
            got = pre.featurize_batch(w.cuda(), normalize=True).cpu()
            ref = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**BASE), **flags)
            assert got.shape == ref.shape == (3, pre.get_num_features(), n // 160 + 1)
            nbase = got.shape[1] - (4 if flags.get("use_spectral_contrast") else 0)
            mel, rel = _errors(got[:, :nbase], ref[:, :nbase], 64)
            cerr = (got[:, nbase:] - ref[:, nbase:]).abs().max().item() if nbase < got.shape[1] else 0.0
            assert mel < FEAT_TOL and rel < 2 * FEAT_TOL and cerr < 2e-4, (flags, n, mel, rel, cerr)
        assert torch.equal(pre.featurize_batch(one.cuda(), normalize=True), base)
