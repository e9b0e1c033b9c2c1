"""The full-band one-launch featuriser (featurize_kernel<..., FULL = true>) vs the CPU oracle.

One constructor argument of the reference -- ``f_max=8000`` (torchaudio's own default when f_max is omitted), ``n_mels=80``,
``n_mfcc=20`` (``/root/reference/src/preprocessing.py:32-51, 94-127``) -- used to drop the shipped STFT geometry from the tuned
kernel to the generic kernel chain.  The full-band instantiations form all 257 bins, keep the filterbank as a CSR table in LDS
and take n_mels / n_mfcc at run time; the stem of the residual net stays fused for the 90-row layout."""
import numpy as np
import pytest
import torch

import cough_detector_amd as cda
from oracle import featurizer as ofeat, resnet as ores
from parity import FEAT_TOL, SHIPPED, edge_clips, realistic_state_dict, synth_batch

pytestmark = pytest.mark.gpu

BASE = dict(sample_rate=16000, n_mels=64, n_fft=512, hop_length=160, win_length=400, f_min=100.0, f_max=4000.0, n_mfcc=13)
FULLBAND = {
    "fmax8k_64mel": dict(f_max=8000.0),
    "mel80_mfcc20": dict(n_mels=80, n_mfcc=20, f_max=8000.0),
    "mel128_fmin0": dict(n_mels=128, f_min=0.0, f_max=8000.0),
    "mel40_mfcc20": dict(n_mels=40, n_mfcc=20, f_max=7000.0),
    "mel24_wide_bands": dict(n_mels=24, n_mfcc=12, f_min=50.0, f_max=8000.0),
    "fmax5k_wide_taps": dict(n_mels=32, f_max=5000.0),
    "mel2": dict(n_mels=2, n_mfcc=2, f_max=8000.0),
    "mel72_idle_lanes_in_the_second_pass": dict(n_mels=72, n_mfcc=16, f_max=8000.0),      # 8 left-over bands: (band, frame) lanes
    "mel100": dict(n_mels=100, n_mfcc=9, f_min=40.0, f_max=7800.0),                       # 36 left-over bands: four frames per lane
}


def _geo(over):
    g = dict(BASE)
    g.update(over)
    return g


def _errors(got, ref, n_mels):
    got, ref = got.detach().cpu().float(), ref.detach().cpu().float()
    mel = (got[..., :n_mels, :] - ref[..., :n_mels, :]).abs().max().item()
    d = (got[..., n_mels:, :] - ref[..., n_mels:, :]).abs()
    return mel, ((d / ref[..., n_mels:, :].abs().clamp(min=1.0)).max().item() if d.numel() else 0.0)


def test_which_constructor_calls_land_on_which_kernels():
    def path(**kw):
        return cda.AudioPreprocessor(device="cuda", **{**SHIPPED, **kw}).kernel_path()
    assert path() == "tuned" and path(f_max=3000.0) == "tuned"
    for over in FULLBAND.values():
        assert path(**over) == "tuned_fullband", over
    assert path(f_max=8000.0, use_pcen=True, use_pre_emphasis=True, use_delta_delta=True) == "tuned_fullband"
    assert path(f_max=8000.0, use_spectral_contrast=True, n_contrast_bands=4) == "tuned_fullband"
    assert path(f_max=8000.0, use_mfcc=False, n_mfcc=40) == "tuned_fullband"          # n_mfcc unused without MFCC rows
    # odd band counts / more than 20 MFCCs / PCEN off 64 bands: the run-time-geometry kernel; outside: other n_fft
    assert path(n_mels=63, f_max=8000.0) == "tuned_geometry"      # odd band counts: the run-time-geometry kernel (element-wise stores)
    assert path(n_mels=80, n_mfcc=21, f_max=8000.0) == "tuned_geometry" and path(n_mels=128, n_mfcc=40, f_max=8000.0) == "tuned_geometry"
    assert path(n_mfcc=42) == "generic"                           # 42 x 101 MFCC values do not fit the 16 640-byte scratch
    assert path(n_mels=80, f_max=8000.0, use_pcen=True) == "tuned_geometry"   # PCEN off 64 bands: the run-time-geometry kernel
    # n_fft = 512 with another hop / window / sample rate / segment of <= 128 frames: the full-band kernel with a run-time geometry
    assert path(hop_length=200) == "tuned_geometry" and path(segment_duration=0.5) == "tuned_geometry"
    assert path(sample_rate=22050, f_max=8000.0, hop_length=220, win_length=441) == "tuned_geometry"
    assert path(hop_length=128, win_length=512) == "tuned_geometry" and path(win_length=37, hop_length=77, n_mels=20, segment_duration=0.5) == "tuned_geometry"
    assert path(hop_length=200, use_spectral_contrast=True, n_contrast_bands=4) == "tuned_geometry"   # (contrast rows: generic kernels)
    assert path(hop_length=100) == "tuned_geometry" and path(segment_duration=2.0) == "tuned_geometry"      # 161 / 201 frames
    assert path(hop_length=600) == "generic" and path(segment_duration=5.0) == "generic"           # hop > 512; 501 frames x 64 bands
    assert path(n_fft=400) == "generic"


@pytest.mark.parametrize("name", sorted(FULLBAND))
def test_fullband_geometry_against_oracle(name):
    g = _geo(FULLBAND[name])
    pre = cda.AudioPreprocessor(device="cuda", **g, **SHIPPED)
    assert pre.kernel_path() == "tuned_fullband"
    edges = edge_clips()
    w = torch.cat([synth_batch(500, 20, peak_normalize=False), torch.from_numpy(np.stack([edges[k] for k in sorted(edges)]))])
    kw = ofeat.geometry_kwargs(**g)
    nm = g["n_mels"]
    hard = [20 + i for i, k in enumerate(sorted(edges)) if k in ("dc", "ramp")]   # ill-conditioned in float32 (tests/parity.py)
    keep = [i for i in range(w.shape[0]) if i not in hard]
    for normalize in (False, True):
        got = pre.featurize_batch(w.cuda(), normalize=normalize).cpu()
        ref = ofeat.extract_features_batch(w, normalize_first=normalize, **kw)
        assert got.shape == ref.shape == (w.shape[0], nm + 2 * g["n_mfcc"], 101)
        mel, rel = _errors(got[keep], ref[keep], nm)
        print(f"{name} normalize={normalize}: mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
        assert torch.isfinite(got).all() and mel < FEAT_TOL and rel < FEAT_TOL
        melh, relh = _errors(got[hard], ref[hard], nm)
        assert melh < 2e-2 and relh < 2e-2          # bounded as for the shipped kernel (every mel value is rounding noise there)
        zi = 20 + sorted(edges).index("zeros")
        assert torch.all(got[zi, :nm] == 0)
        # batch invariance: every reduction is per clip
        one = pre.featurize_batch(w[5:6].cuda(), normalize=normalize).cpu()
        assert torch.equal(one[0], got[5])


@pytest.mark.parametrize("flags", [dict(use_pre_emphasis=True), dict(use_delta_delta=True), dict(use_mfcc=False),
                                   dict(use_pre_emphasis=True, use_delta_delta=True),
                                   dict(use_spectral_contrast=True, n_contrast_bands=4, use_delta_delta=True)],
                         ids=["preemph", "dd", "no_mfcc", "preemph_dd", "contrast4_dd"])
@pytest.mark.parametrize("name", ["fmax8k_64mel", "mel80_mfcc20"])
def test_fullband_flags(name, flags):
    g = _geo(FULLBAND[name])
    kw = {**SHIPPED, **flags}
    pre = cda.AudioPreprocessor(device="cuda", **g, **kw)
    assert pre.kernel_path() == "tuned_fullband"
    w = synth_batch(540, 12, peak_normalize=False)
    got = pre.featurize_batch(w.cuda(), normalize=True).cpu()
    ref = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**g), **kw)
    assert got.shape == ref.shape == (12, pre.get_num_features(), 101)
    nm = g["n_mels"]
    nbase = nm + ((2 + bool(kw["use_delta_delta"])) * g["n_mfcc"] if kw.get("use_mfcc", True) else 0)
    mel, rel = _errors(got[:, :nbase], ref[:, :nbase], nm)
    print(f"{name} {flags}: mel abs {mel:.2e}, rest rel {rel:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL
    if kw["use_spectral_contrast"]:
        assert (got[:, nbase:] - ref[:, nbase:]).abs().max().item() < 2e-4


def test_fullband_pcen_at_64_bands():
    g = _geo(FULLBAND["fmax8k_64mel"])
    kw = {**SHIPPED, "use_pcen": True, "use_pre_emphasis": True, "use_delta_delta": True}
    pre = cda.AudioPreprocessor(device="cuda", **g, **kw)
    assert pre.kernel_path() == "tuned_fullband"
    w = synth_batch(560, 12, peak_normalize=False)
    got = pre.featurize_batch(w.cuda(), normalize=True).cpu()
    ref = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**g), **kw)
    mel, rel = _errors(got, ref, 64)
    print(f"PCEN full band: mel abs {mel:.2e}, rest rel {rel:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL


@pytest.mark.parametrize("dtype", ["bf16x3", "fp32"])
def test_fullband_stem_stays_fused_and_equals_featurise_then_classify(dtype):
    """f_max = 8000 keeps the 90-row layout: with the split-bf16 net the stem runs inside the full-band featurise kernel
    (no feature image in HBM); logits are bit-identical to featurise -> classify and within 1e-3 of the CPU oracle."""
    g = _geo(FULLBAND["fmax8k_64mel"])
    sd = realistic_state_dict(13)
    pre = cda.AudioPreprocessor(device="cuda", **g, **SHIPPED)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype)
    model.load_state_dict(sd)
    model.eval()
    pipe = cda.CoughPipeline(pre, model)
    w = synth_batch(580, 40, peak_normalize=False)
    logits, feats = pipe(w.cuda(), normalize=True, return_features=True)
    ref_f = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**g))
    mel, rel = _errors(feats, ref_f, 64)
    assert mel < FEAT_TOL and rel < FEAT_TOL
    ref_l = ores.forward(ref_f[:, None], sd)
    err = (logits.cpu() - ref_l).abs().max().item()
    print(f"full-band pipeline {dtype}: logits max abs err {err:.2e}")
    assert err < 1e-3 and torch.equal(logits.cpu().argmax(1), ref_l.argmax(1))
    two_step = model(pre.featurize_batch(w.cuda(), normalize=True)[:, None])
    assert torch.equal(logits, two_step)
    assert torch.equal(pipe(w.cuda(), normalize=True), logits)                  # without the materialised features too


def test_fullband_full_size_batch_properties():
    """B = 4096 through size-independent properties: batch invariance, mel rows in [0, 1], z-scored rows mean 0 / unbiased
    std 1 per clip, delta rows = central difference of the stored MFCC rows."""
    g = _geo(FULLBAND["mel80_mfcc20"])
    pre = cda.AudioPreprocessor(device="cuda", **g, **SHIPPED)
    from cough_detector_amd import synth
    w = synth.device_clips(0, 4096)
    f = pre.featurize_batch(w, normalize=True)
    assert f.shape == (4096, 120, 101) and torch.isfinite(f).all()
    small = pre.featurize_batch(w[1000:1006], normalize=True)
    assert torch.equal(small, f[1000:1006])
    mel, mf, dl = f[:, :80], f[:, 80:100], f[:, 100:120]
    assert mel.min().item() >= 0.0 and mel.max().item() <= 1.0
    flat = mf.reshape(4096, -1)
    assert flat.mean(dim=1).abs().max().item() < 1e-4 and (flat.std(dim=1) - 1).abs().max().item() < 1e-4
    pad = torch.nn.functional.pad(mf, (1, 1), mode="replicate")
    assert torch.equal(dl, (pad[:, :, 2:] - pad[:, :, :-2]) / 2)


@pytest.mark.parametrize("sr,seconds,over", [(8000, 2.0, dict(f_max=4000.0)), (32000, 0.5, dict(f_max=8000.0, n_mels=80, n_mfcc=20)),
                                             (8000, 2.0, dict(f_max=1900.0))],
                         ids=["8kHz_2s", "32kHz_half_second_80mel", "8kHz_2s_narrow_bank"])
def test_same_stft_at_another_sample_rate(sr, seconds, over):
    """16 000 samples / hop 160 / window 400 at another sample rate is the shipped STFT with another filterbank (which arrives as a
    table): the one-launch kernels with the fused stem serve it, contrast + centroid rows included (the centroid is a ratio:
    sum(f_k |X_k|) / sum(|X_k|) / (sample_rate / 2) does not depend on the rate)."""
    from test_oracle_featurizer import geometry_clip
    g = _geo({"sample_rate": sr, **over})
    flags = {**SHIPPED, "use_spectral_contrast": True, "n_contrast_bands": 3, "use_delta_delta": True}
    pre = cda.AudioPreprocessor(device="cuda", segment_duration=seconds, **g, **flags)
    assert pre.segment_samples == 16000 and pre.kernel_path() == ("tuned" if over["f_max"] < sr / 4 else "tuned_fullband")
    w = torch.from_numpy(np.stack([geometry_clip(s, 16000) for s in range(8)]))
    got = pre.featurize_batch(w.cuda(), normalize=True).cpu()
    ref = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**g), **flags)
    nm = g["n_mels"]
    nbase = nm + 3 * g["n_mfcc"]
    assert got.shape == ref.shape == (8, nbase + 4, 101)
    mel, rel = _errors(got[:, :nbase], ref[:, :nbase], nm)
    cerr = (got[:, nbase:] - ref[:, nbase:]).abs().max().item()
    print(f"{sr} Hz x {seconds} s [{pre.kernel_path()}]: mel abs {mel:.2e}, rest rel {rel:.2e}, contrast rows abs {cerr:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL and cerr < 2e-4
