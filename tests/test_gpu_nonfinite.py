"""Non-finite and extreme-peak SAMPLES: the HIP path must give what the reference gives.

One NaN / Inf sample makes the reference's whole feature image NaN -- ``normalize`` leaves a clip with a NaN maximum alone
(``/root/reference/src/preprocessing.py:199-212``: ``NaN > 0`` is False; an infinite maximum divides everything to 0 / NaN),
the frames over the bad sample are NaN in every bin, ``AmplitudeToDB``'s per-clip ``amax`` is NaN and its floor poisons every
cell (:405-410), the z-score and the deltas follow (:428) -- hence NaN logits, an engine that does not fire and a NaN that
stays in the smoothing deque for ``smoothing_window`` windows (``/root/reference/src/inference.py:184-189, 220-223``).
A clip whose peak is denormal or huge is, with ``normalize``, an ordinary clip in the reference (it rescales first)."""
import numpy as np
import pytest
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from oracle import engine as oengine, featurizer as ofeat, resnet as ores
from parity import FEAT_TOL, SHIPPED, feature_errors, realistic_state_dict, synth_batch

pytestmark = pytest.mark.gpu

NAN, INF = float("nan"), float("inf")
BAD_SAMPLES = {"nan_first": (0, NAN), "nan_middle": (8000, NAN), "nan_last": (15999, NAN), "plus_inf": (777, INF),
               "minus_inf": (12345, -INF)}
# the one-launch kernel (shipped sparse bank, full-band bank, run-time STFT geometry) and two geometries of the generic kernel chain
GEOMETRIES = {"tuned": {}, "tuned_fullband_fmax8k": dict(f_max=8000.0), "tuned_geometry_hop200_40mel": dict(hop_length=200, n_mels=40),
              "generic_nfft400": dict(n_fft=400), "generic_42_mfcc": dict(n_mfcc=42), "tuned_geometry_hop100_161_frames": dict(hop_length=100)}


def _oracle(w, normalize, geo, **flags):
    kw = dict(flags)
    if geo:
        g = dict(sample_rate=16000, n_mels=64, n_fft=512, hop_length=160, win_length=400, f_min=100.0, f_max=4000.0, n_mfcc=13)
        g.update(geo)
        kw.update(ofeat.geometry_kwargs(**g))
    return ofeat.extract_features_batch(w, normalize_first=normalize, **kw)


def _same_with_nans(got, ref, tol=FEAT_TOL):
    got, ref = got.detach().cpu(), ref.detach().cpu()
    assert got.shape == ref.shape
    gn, rn = torch.isnan(got), torch.isnan(ref)
    assert torch.equal(gn, rn), f"NaN cells differ: {int(gn.sum())} here vs {int(rn.sum())} in the reference path"
    d = (got[~rn] - ref[~rn]).abs() / ref[~rn].abs().clamp(min=1.0)
    assert d.numel() == 0 or d.max().item() < tol, d.max().item()


@pytest.mark.parametrize("geo", list(GEOMETRIES))
@pytest.mark.parametrize("normalize", [False, True])
def test_one_bad_sample_makes_the_whole_image_nan_and_leaves_the_neighbours_alone(geo, normalize):
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED, **GEOMETRIES[geo])
    assert pre.kernel_path() == next(p for p in ("generic", "tuned_fullband", "tuned_geometry", "tuned") if geo.startswith(p))
    clean = synth_batch(300, len(BAD_SAMPLES) + 2, peak_normalize=False)
    w = clean.clone()
    for i, (pos, val) in enumerate(BAD_SAMPLES.values()):
        w[i + 1, pos] = val                                   # clips 0 and -1 stay clean
    got = pre.featurize_batch(w.cuda(), normalize=normalize).cpu()
    ref = _oracle(w, normalize, GEOMETRIES[geo])
    for i in range(1, len(BAD_SAMPLES) + 1):
        assert torch.isnan(ref[i]).all()                      # what the reference path gives (the premise of this test)
        assert torch.isnan(got[i]).all(), f"{list(BAD_SAMPLES)[i - 1]}: {int(torch.isnan(got[i]).sum())} of {got[i].numel()} NaN"
    _same_with_nans(got, ref)
    alone = pre.featurize_batch(clean.cuda(), normalize=normalize).cpu()
    assert torch.equal(got[0], alone[0]) and torch.equal(got[-1], alone[-1])      # per-clip: the neighbours are bit-identical


@pytest.mark.parametrize("geo", ["tuned", "tuned_fullband_fmax8k", "tuned_geometry_hop200_40mel", "generic_nfft400"])
@pytest.mark.parametrize("flags", [dict(use_delta_delta=True), dict(use_pcen=True, use_pre_emphasis=True),
                                   dict(use_mfcc=False), dict(use_spectral_contrast=True, n_contrast_bands=4),
                                   dict(use_pre_emphasis=True, use_delta_delta=True)],
                         ids=["dd", "pcen_preemph", "no_mfcc", "contrast4", "preemph_dd"])
def test_every_flag_branch_follows_the_rule(geo, flags):
    kw = {**SHIPPED, **flags}
    pre = cda.AudioPreprocessor(device="cuda", **kw, **GEOMETRIES[geo])
    w = synth_batch(320, 4, peak_normalize=False)
    w[1, 4000] = NAN
    w[2, 15998] = INF
    okw = {k: v for k, v in kw.items()}
    for normalize in (False, True):
        got = pre.featurize_batch(w.cuda(), normalize=normalize).cpu()
        ref = _oracle(w, normalize, GEOMETRIES[geo], **okw)
        assert torch.isnan(ref[1]).all() and torch.isnan(ref[2]).all() and not torch.isnan(ref[0, :40]).any()   # (contrast rows may be NaN by construction)
        _same_with_nans(got, ref)


@pytest.mark.parametrize("geo", list(GEOMETRIES))
def test_denormal_and_huge_peaks_are_ordinary_clips_under_normalize(geo):
    """The reference divides by the peak first, so a clip at 1e-42 (denormal), 1e-30, 1e+25 or 1e+30 full scale gives the features of
    the same clip at unit peak (up to the few mantissa bits a denormal sample keeps); the HIP kernels apply the normalisation
    after the transform and re-run such a clip with power-of-two-scaled window taps (featurize.hip) / divide exactly (generic)."""
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED, **GEOMETRIES[geo])
    base = synth_batch(340, 1, peak_normalize=True)[0]
    scales = [1e-42, 1e-38, 1e-30, 1e-19, 1.0, 1e19, 1e25, 1e30]
    w = torch.stack([(base.double() * s).float() for s in scales])
    got = pre.featurize_batch(w.cuda(), normalize=True).cpu()
    ref = _oracle(w, True, GEOMETRIES[geo])
    assert not torch.isnan(ref).any()
    mel, rel = feature_errors(got, ref) if not GEOMETRIES[geo].get("n_mels") else \
        ((got - ref).abs().max().item(), 0.0)
    print(f"{geo}: extreme peaks under normalize: mel {mel:.2e} rest {rel:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL
    # without normalize the reference overflows / underflows in float32 exactly as the kernels do: huge -> NaN image, tiny -> silence
    got = pre.featurize_batch(w.cuda(), normalize=False).cpu()
    ref = _oracle(w, False, GEOMETRIES[geo])
    assert torch.isnan(ref[-1]).all() and not torch.isnan(ref[0]).any()
    _same_with_nans(got, ref, tol=2e-4)


@pytest.mark.parametrize("dtype", ["bf16x3", "fp32", "bf16_approx"])
def test_pipeline_gives_nan_logits_for_the_bad_clip_only(dtype):
    sd = realistic_state_dict(11)
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype)
    model.load_state_dict(sd)
    model.eval()
    pipe = cda.CoughPipeline(pre, model)
    clean = synth_batch(360, 6, peak_normalize=False)
    w = clean.clone()
    w[2, 9000] = NAN
    w[4, 0] = -INF
    logits = pipe(w.cuda(), normalize=True).cpu()
    preds, probs = pipe.predict(w.cuda(), normalize=True)
    ref = ores.forward(_oracle(w, True, {})[:, None], sd)
    assert torch.isnan(ref[2]).all() and torch.isnan(ref[4]).all()        # torch propagates the NaN image to both logits
    for i in (2, 4):
        assert torch.isnan(logits[i]).all() and torch.isnan(probs[i].cpu()).all() and int(preds[i]) == 0
    good = [0, 1, 3, 5]
    assert not torch.isnan(logits[good]).any()
    if dtype != "bf16_approx":
        assert (logits[good] - ref[good]).abs().max().item() < 1e-3
    alone = pipe(clean.cuda(), normalize=True).cpu()
    assert torch.equal(logits[good], alone[good])                          # the bad clips do not touch their neighbours
    # the materialised features of the fused pipeline are the reference's NaN image too
    _, feats = pipe(w.cuda(), normalize=True, return_features=True)
    assert torch.isnan(feats[2]).all() and torch.isnan(feats[4]).all() and not torch.isnan(feats[good]).any()


def test_engine_keeps_a_nan_in_its_smoothing_history_like_the_reference(tmp_path):
    """A NaN sample in the stream: every window over it has a NaN probability, the smoothed mean is NaN for smoothing_window
    windows after the last of them, `NaN >= threshold` is False, so the engine stays silent exactly as long as the reference's
    (src/inference.py:220-223)."""
    from test_gpu_engine import make_checkpoint
    sd = realistic_state_dict(5)
    sd["fc.2.bias"] = sd["fc.2.bias"] + torch.tensor([0.0, 1.5])            # fires readily on clean windows
    path = make_checkpoint(tmp_path, sd)
    now = {"t": 0.0}
    eng = cda.CoughDetectorInference(path, device="auto", confidence_threshold=0.5, smoothing_window=3,
                                     debounce_seconds=0.5, verbose=False, clock=lambda: now["t"])
    ref = oengine.EngineOracle(sd, 0.5, 3, 0.5, clock=lambda: now["t"])
    stream = synth.make_stream(21, 6.0).copy()
    stream[40000] = NAN                                                     # 2.5 s: windows starting in (1.5 s, 2.5 s]
    events, ref_events = [], []
    for i in range(0, len(stream), 1600):
        now["t"] = (i + 1600) / 16000.0
        a = eng.process_audio_chunk(stream[i:i + 1600])
        b = ref.process_audio_chunk(stream[i:i + 1600])
        events.append(a is not None)
        ref_events.append(b is not None)
    got, want = np.array(eng.window_probs), np.array(ref.window_probs)
    assert len(got) == len(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want).sum() == 4
    ok = ~np.isnan(want)
    assert np.abs(got[ok] - want[ok]).max() < 1e-3
    assert events == ref_events and any(events)


@pytest.mark.parametrize("geo", ["tuned", "tuned_fullband_fmax8k", "generic_42_mfcc", "tuned_geometry_hop100_161_frames"])
def test_contrast_rows_of_extreme_peak_clips_under_normalize(geo):
    """The spectral-contrast rows come from a second STFT of the un-emphasised signal; with `normalize` the reference has divided
    by the peak first (:199-212, :476-478), so a denormal or huge clip gives the rows of the same clip at unit peak.  The tuned
    path's power pass scales such a clip's samples by a power of two (the featurise kernel leaves every clip's peak)."""
    flags = {**SHIPPED, "use_spectral_contrast": True, "n_contrast_bands": 4}
    pre = cda.AudioPreprocessor(device="cuda", **flags, **GEOMETRIES[geo])
    base = synth_batch(345, 1, peak_normalize=True)[0]
    scales = [1e-42, 1e-30, 1e-19, 1e-3, 1.0, 1e19, 1e30]
    w = torch.stack([(base.double() * s).float() for s in scales])
    got = pre.featurize_batch(w.cuda(), normalize=True).cpu()
    ref = _oracle(w, True, GEOMETRIES[geo], **flags)
    assert got.shape == ref.shape and torch.isfinite(ref).all()
    nb = got.shape[1] - 5
    err_rows = (got[:, nb:] - ref[:, nb:]).abs().amax(dim=(1, 2))
    err_front = (got[:, :nb] - ref[:, :nb]).abs().max().item()
    print(f"{geo}: contrast rows of extreme-peak clips: per clip {[float('%.1e' % e) for e in err_rows]}, rows in front {err_front:.1e}")
    assert err_front < 1e-4
    assert err_rows[1:].max().item() < 5e-5          # 1e-30 .. 1e30
    assert err_rows[0].item() < 5e-2                 # 1e-42: the samples themselves keep ~9 bits (712 denormal steps)
