"""K1 parity: HIP featuriser (through the C-ABI) vs the CPU oracle, goldens and size-independent properties."""
import numpy as np
import pytest
import torch

import cough_detector_amd as cda
from oracle import dft64, featurizer as ofeat
from parity import FEAT_TOL, SHIPPED, edge_clips, feature_errors, strict_feature_error, synth_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pre():
    return cda.AudioPreprocessor(device="cuda", **SHIPPED)


def test_reference_call_shape_single_clip(pre):
    w = synth_batch(0, 1)
    f = pre.extract_features(w)                      # (1, N) -> (1, 90, 101), the reference contract
    assert f.shape == (1, 90, 101) and f.dtype == torch.float32 and f.is_cuda
    mel, rel = feature_errors(f, ofeat.extract_features(w))
    assert mel < FEAT_TOL and rel < FEAT_TOL
    cpu_pre = cda.AudioPreprocessor(device="cpu", **SHIPPED)
    assert not cpu_pre.extract_features(w).is_cuda   # device="cpu" returns host tensors like the reference


def test_golden_fixture(pre, features_golden):
    w = synth_batch(0, len(features_golden["seeds"]))
    f = pre.extract_features(w.cuda())
    mel, rel = feature_errors(f, torch.from_numpy(features_golden["features"]))
    print(f"golden: mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL


def test_synthetic_mixture_against_oracle(pre):
    w = synth_batch(100, 96)
    f = pre.extract_features(w.cuda())
    ref = ofeat.extract_features_batch(w)
    mel, rel = feature_errors(f, ref)
    print(f"synthetic x96: mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
    assert mel < 2e-5 and rel < 2e-5          # measured 3e-6 / 2e-6; spec bound is FEAT_TOL = 1e-4


def test_strict_survey_metric_is_reported_next_to_the_oracles_own_float64_error(pre, features_golden):
    """SURVEY 8d defines the relative error of the z-scored rows with max(|ref|, 1e-3); the suite's
    metric uses max(|ref|, 1).  Both are computed here on the 32 golden clips, together with the float32 CPU oracle's
    own distance from the independent float64 re-derivation in the same strict metric: the HIP path must sit within 3x
    of what the float32 oracle itself achieves (at a zero crossing of a unit-variance row a 4e-6 absolute error reads
    as 4e-3 'relative'), and within the strict 1e-4 wherever |ref| >= 0.05."""
    w = synth_batch(0, len(features_golden["seeds"]))
    got = pre.extract_features(w.cuda()).cpu()
    ref = torch.from_numpy(features_golden["features"])
    truth = torch.from_numpy(np.stack([dft64.features(w[i].numpy()) for i in range(w.shape[0])]))
    hip_strict, oracle_strict = strict_feature_error(got, ref), strict_feature_error(ref, truth)
    hip_truth = strict_feature_error(got, truth)
    _, loose = feature_errors(got, ref)
    d = (got[:, 64:].double() - ref[:, 64:].double()).abs()
    big = ref[:, 64:].abs() >= 0.05
    rel_big = (d[big] / ref[:, 64:].double().abs()[big]).max().item()
    print(f"z-scored rows, 32 goldens: |a-b|/max(|b|,1) = {loose:.2e};  SURVEY 8d strict |a-b|/max(|b|,1e-3): HIP vs oracle "
          f"{hip_strict:.2e}, oracle(f32) vs float64 {oracle_strict:.2e}, HIP vs float64 {hip_truth:.2e};  max abs {d.max():.2e};  "
          f"relative where |ref| >= 0.05: {rel_big:.2e}")
    assert d.max().item() < 2e-5 and rel_big < FEAT_TOL
    assert hip_truth < 3 * max(oracle_strict, FEAT_TOL) and hip_strict < 3 * max(oracle_strict, FEAT_TOL)


@pytest.mark.parametrize("name", sorted(edge_clips().keys()))
def test_edge_cases(pre, name):
    w = torch.from_numpy(edge_clips()[name])[None]
    f = pre.extract_features(w.cuda())
    assert torch.isfinite(f).all()
    ref = ofeat.extract_features(w)
    mel, rel = feature_errors(f, ref)
    # conditioning: on inputs whose energy sits entirely in bins 0-1 (dc, ramp) the mel bands hold only
    # f32 rounding noise 13 orders below the peak, and the f32 CPU path itself is 7e-4 .. 1.5e-3 away
    # from float64 truth -- no two f32 implementations agree there.  Bound by the oracle's own f64 error.
    truth = torch.from_numpy(dft64.features(edge_clips()[name])).float()[None]
    ref_mel, ref_rel = feature_errors(ref, truth)
    print(f"{name}: mel abs {mel:.2e}, mfcc/delta rel {rel:.2e} (oracle vs f64: {ref_mel:.2e}, {ref_rel:.2e})")
    assert mel < max(FEAT_TOL, 3 * ref_mel) and rel < max(FEAT_TOL, 3 * ref_rel)
    if name not in ("dc", "ramp"):
        assert mel < FEAT_TOL and rel < FEAT_TOL
    if name == "zeros":
        assert torch.all(f[0, :64] == 0)


def test_fused_normalize_matches_reference_order(pre):
    raw = synth_batch(200, 24, peak_normalize=False) * 0.37
    raw[5] = 0.0                                      # all-zero clip: normalize is a silent no-op
    got = pre.featurize_batch(raw.cuda(), normalize=True)
    ref = ofeat.extract_features_batch(raw, normalize_first=True)
    mel, rel = feature_errors(got, ref)
    assert mel < FEAT_TOL and rel < FEAT_TOL
    scaled = pre.featurize_batch((raw * 7.5).cuda(), normalize=True)     # peak-normalised => scale invariant
    mel, rel = feature_errors(scaled, got)
    assert mel < FEAT_TOL and rel < FEAT_TOL


def test_optional_flags_pre_emphasis_and_delta_delta():
    w = synth_batch(300, 8)
    for kw in (dict(use_pre_emphasis=True), dict(use_delta_delta=True), dict(use_pre_emphasis=True, use_delta_delta=True)):
        flags = {**SHIPPED, **kw}
        p = cda.AudioPreprocessor(device="cuda", **flags)
        f = p.extract_features(w.cuda())
        ref = ofeat.extract_features_batch(w, use_pre_emphasis=flags["use_pre_emphasis"],
                                           use_delta_delta=flags["use_delta_delta"])
        assert f.shape == ref.shape
        mel, rel = feature_errors(f, ref)
        assert mel < FEAT_TOL and rel < FEAT_TOL, kw


def test_realtime_add_audio_matches_oracle():
    rt = cda.RealtimePreprocessor(window_duration=1.0, hop_duration=0.25, device="cuda", **SHIPPED)
    ow = ofeat.RealtimeWindowerOracle(1.0, 0.25)
    from cough_detector_amd import synth
    stream = torch.from_numpy(synth.make_stream(5, 2.5))
    n = 0
    for i in range(0, stream.numel(), 1600):
        got, want = rt.add_audio(stream[i:i + 1600]), ow.add_audio(stream[i:i + 1600])
        assert len(got) == len(want)
        for g, r in zip(got, want):
            assert g.shape == (1, 90, 101)
            mel, rel = feature_errors(g, r)
            assert mel < FEAT_TOL and rel < FEAT_TOL
            n += 1
    assert n == 7


def test_empty_batch_and_strided_input(pre):
    assert pre.featurize_batch(torch.zeros(0, 16000, device="cuda")).shape == (0, 90, 101)
    big = torch.zeros(5, 16004, device="cuda")
    w = synth_batch(40, 5)
    big[:, :16000] = w.cuda()
    view = big[:, :16000]                             # row stride 16004 (multiple of 4), no copy needed
    mel, rel = feature_errors(pre.featurize_batch(view), ofeat.extract_features_batch(w))
    assert mel < FEAT_TOL and rel < FEAT_TOL


def test_full_size_batch_properties(pre):
    """BASELINE configs[1]: B = 4096.  Oracle on a sample + size-independent properties on everything."""
    B = 4096
    w = synth_batch(1000, B)
    wd = w.cuda()
    f = pre.featurize_batch(wd)
    assert f.shape == (B, 90, 101) and torch.isfinite(f).all()
    # batch invariance (per-clip reductions, no cross-clip leakage): bit-exact vs small launches
    idx = torch.tensor([0, 1, 255, 256, 2047, 4095])
    assert torch.equal(pre.featurize_batch(wd[idx]), f[idx])
    assert torch.equal(pre.featurize_batch(wd), f)    # deterministic
    # oracle on a sample
    sample = torch.arange(0, B, 67)
    mel, rel = feature_errors(f[sample], ofeat.extract_features_batch(w[sample]))
    print(f"B=4096 sample: mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL
    # properties: mel rows in [0,1]; z-scored MFCC has mean 0 / unbiased std 1 per clip; delta rows are
    # exactly the central difference of the stored MFCC rows
    assert f[:, :64].min() >= 0 and f[:, :64].max() <= 1
    z = f[:, 64:77].reshape(B, -1)
    assert z.mean(dim=1).abs().max() < 1e-4 and (z.std(dim=1) - 1).abs().max() < 1e-4
    zz = f[:, 64:77]
    zp = torch.cat([zz[:, :, :1], zz, zz[:, :, -1:]], dim=2)
    assert torch.equal((zp[:, :, 2:] - zp[:, :, :-2]) / 2, f[:, 77:90])


def test_pcen_branch():
    """8f 'next' row: use_pcen=True (a constructor default of the reference), mel rows = min-max PCEN."""
    w = synth_batch(400, 16)
    raw = synth_batch(500, 8, peak_normalize=False) * 0.3
    flags = {**SHIPPED, "use_pcen": True}
    p = cda.AudioPreprocessor(device="cuda", **flags)
    f = p.extract_features(w.cuda())
    ref = ofeat.extract_features_batch(w, use_pcen=True)
    mel, rel = feature_errors(f, ref)
    print(f"pcen: mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL
    assert f[:, :64].min() >= 0 and f[:, :64].max() <= 1
    got = p.featurize_batch(raw.cuda(), normalize=True)
    ref = ofeat.extract_features_batch(raw, normalize_first=True, use_pcen=True)
    mel, rel = feature_errors(got, ref)
    assert mel < FEAT_TOL and rel < FEAT_TOL


@pytest.mark.parametrize("orig_sr,channels,seconds", [(44100, 2, 1.3), (22050, 1, 0.5), (48000, 1, 1.0), (8000, 2, 1.7)])
def test_process_front_end_resample_mono_normalize_pad(orig_sr, channels, seconds):
    """8f 'next' row: process() = resample -> mono -> normalize -> pad/trim -> features (preprocessing.py:491-517)."""
    g = torch.Generator().manual_seed(orig_sr + channels)
    n = int(orig_sr * seconds)
    t = torch.arange(n) / orig_sr
    w = torch.stack([0.3 * torch.sin(2 * torch.pi * (300 + 250 * c) * t) + 0.05 * torch.randn(n, generator=g)
                     for c in range(channels)])
    p = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    r = p.resample(w, orig_sr)
    r_ref = ofeat.resample(w, orig_sr)
    assert r.shape == r_ref.shape
    assert (r.cpu() - r_ref).abs().max() < 2e-6
    f = p.process(w, orig_sr)
    ref = ofeat.process(w, orig_sr)
    assert f.shape == ref.shape == (1, 90, 101)
    mel, rel = feature_errors(f, ref)
    print(f"process {orig_sr} Hz x{channels}: mel abs {mel:.2e}, mfcc/delta rel {rel:.2e}")
    assert mel < FEAT_TOL and rel < FEAT_TOL


def _spec_err(got, ref, floor_rel):
    """max over elements of |got - ref| / (|ref| + floor_rel * max|ref| of the clip): relative, above the fp32
    FFT's own noise floor (spectra span > 12 decades inside one clip).  An fp32 FFT -- torch's included -- carries
    ~1e-7 * A_max of absolute amplitude noise in every bin.  For magnitudes that is 1e-7 / floor_rel of the
    denominator at worst: floor_rel = 1e-2 leaves a 10x margin under 1e-4.  For powers the error of a bin of
    amplitude a is 2 * a * 1e-7 * A_max, which relative to a^2 + floor_rel * A_max^2 peaks at a^2 = floor at
    1e-7 / sqrt(floor_rel): floor_rel = 1e-5 puts that at 3e-5."""
    floor = ref.amax(dim=(-1, -2), keepdim=True) * floor_rel + 1e-30
    return float(((got - ref).abs() / (ref.abs() + floor)).max())


@pytest.mark.parametrize("power,full_window", [(2.0, False), (1.0, False), (2.0, True), (1.0, True)])
def test_stft_stage_against_oracle(pre, power, full_window):
    """cough_spectrogram = T.Spectrogram(512, 400, 160, power) / SpectralCentroid's internal Hann(512) STFT."""
    names = sorted(edge_clips().keys())
    w = torch.cat([synth_batch(300, 24), torch.stack([torch.from_numpy(edge_clips()[n]).float() for n in names])])
    got = pre.spectrogram_batch(w.cuda(), power=power, full_window=full_window).cpu()
    assert got.shape == (w.shape[0], 257, 101)
    ref = ofeat.stft_power(w, win=512 if full_window else 400, power=power)
    err = _spec_err(got, ref, 1e-5 if power == 2.0 else 1e-2)
    print(f"stft power={power} full_window={full_window}: rel err {err:.2e}")
    assert err < FEAT_TOL
    if power == 2.0 and not full_window:      # and against the float64 DFT, all 257 bins
        x = w[3].numpy()
        e64 = _spec_err(got[3].double(), torch.from_numpy(dft64.stft_power(x)), 1e-5)
        assert e64 < FEAT_TOL
    zero_row = names.index("zeros") + 24 if "zeros" in names else None
    if zero_row is not None:
        assert float(got[zero_row].abs().max()) == 0.0


def test_stft_stage_full_batch_parseval(pre):
    """Size-independent check at B = 4096: Parseval per frame.  For a real frame y (windowed, zero outside the
    400 live taps) sum_k c_k |Y_k|^2 = 512 * sum_n y_n^2 with c_0 = c_256 = 1, else 2."""
    B = 4096
    w = synth_batch(0, 64).repeat(B // 64, 1).cuda()
    p = pre.spectrogram_batch(w)                                         # (B, 257, 101)
    c = torch.full((257, 1), 2.0, device="cuda")
    c[0] = c[256] = 1.0
    lhs = (p * c).sum(dim=1)                                             # (B, 101)
    win = torch.zeros(512, device="cuda")
    win[56:456] = ofeat.hann_window(400).cuda()
    padded = torch.nn.functional.pad(w[:, None], (256, 256), mode="reflect")[:, 0]
    frames = padded.unfold(-1, 512, 160) * win                           # (B, 101, 512)
    rhs = 512.0 * (frames.double() ** 2).sum(-1)
    rel = ((lhs.double() - rhs).abs() / (rhs + 1e-12 * rhs.max())).max()
    assert float(rel) < 1e-4
    assert torch.equal(p[:64], p[-64:])                                  # deterministic across workgroups


@pytest.mark.parametrize("b", [1, 255, 300, 513])
def test_stft_stage_persistent_grid_and_output_alignment(pre, b):
    """The STFT kernel is persistent (one workgroup per CU striding over the clips) and writes a clip's image with
    16-byte stores whose alignment follows the OUTPUT address: batches below / around / above the CU count (workgroups
    with 0, 1 and 2 clips), strided waveform rows, and an output pointer at every 4-byte phase -- each must give
    bit-identical spectrograms for the same clip, equal to the oracle's."""
    from cough_detector_amd import _lib
    pool = synth_batch(500, 16)
    ref = ofeat.stft_power(pool, win=400, power=2.0)
    idx = torch.arange(b) % 16
    wav = torch.zeros((b, 16004), device="cuda")
    wav[:, :16000] = pool[idx].cuda()
    base = pre.spectrogram_batch(wav[:, :16000]).cpu()                    # row stride 16004
    assert _spec_err(base[:16], ref[idx[:16]], 1e-5) < FEAT_TOL
    assert torch.equal(base, base[:16][idx] if b >= 16 else base)          # every copy of a clip: the same bits
    lib, img = _lib.load(), 257 * 101
    for phase in (1, 2, 3):
        buf = torch.full((b * img + 8,), float("nan"), device="cuda")
        out = buf[phase:phase + b * img]
        _lib.check(lib.cough_spectrogram(pre._native(), wav.data_ptr(), 16004, out.data_ptr(), b, 0,
                                         torch.cuda.current_stream().cuda_stream), "cough_spectrogram")
        assert torch.equal(out.view(b, 257, 101).cpu(), base), phase
        assert bool(torch.isnan(buf[:phase]).all()) and bool(torch.isnan(buf[phase + b * img:]).all())   # nothing outside


@pytest.mark.parametrize("n_bands", [1, 2, 3, 4])
def test_spectral_contrast_rows_against_oracle(n_bands):
    """use_spectral_contrast=True (preprocessing.py:242-303, :476-480): band contrast + centroid rows, z-scored
    together, appended after the MFCC block; computed from the un-emphasised, optionally normalised signal."""
    flags = dict(use_pcen=False, use_pre_emphasis=True, use_delta_delta=True, use_spectral_contrast=True,
                 n_contrast_bands=n_bands)
    p = cda.AudioPreprocessor(device="cuda", **flags)
    w = synth_batch(500, 20)
    got = p.featurize_batch(w.cuda(), normalize=True).cpu()
    nrow = 64 + 39 + n_bands + 1
    assert got.shape == (20, nrow, 101) and p.get_num_features() == nrow
    ref = ofeat.extract_features_batch(w, normalize_first=True, use_pre_emphasis=True, use_delta_delta=True,
                                       use_spectral_contrast=True, n_contrast_bands=n_bands)
    mel, rel = feature_errors(got[:, :103], ref[:, :103])
    assert mel < FEAT_TOL and rel < FEAT_TOL                       # the rows in front are untouched
    c_got, c_ref = got[:, 103:], ref[:, 103:]
    assert torch.isfinite(c_got).all()
    # rows are z-scored to unit variance over the block; relative to max(|ref|, 0.1) so that the zero crossings of
    # the z-score (|ref| ~ 1e-3, where one ulp of the un-normalised value is already 3e-4 of it) do not set the bar
    err = float(((c_got - c_ref).abs() / c_ref.abs().clamp_min(0.1)).max())
    abs_err = float((c_got - c_ref).abs().max())
    print(f"contrast n_bands={n_bands}: rel err {err:.2e}, abs err {abs_err:.2e}")
    assert err < FEAT_TOL and abs_err < 1e-5
    one = p.extract_spectral_contrast(w[:1])                      # the reference's method name, (1, N) -> (1, n+1, T)
    assert one.shape == (1, n_bands + 1, 101)


def test_reference_default_constructor_reproduces_nan_rows():
    """AudioPreprocessor() -- PCEN, pre-emphasis, delta-delta, 6 contrast bands = 110 rows.  The reference's 7
    contrast rows are NaN for every input (one-bin first band); so are ours, and the 103 rows in front match."""
    with pytest.warns(UserWarning, match="NaN by construction"):
        p = cda.AudioPreprocessor(device="cuda")
    w = synth_batch(600, 6)
    got = p.extract_features(w.cuda()).cpu()
    ref = ofeat.extract_features_batch(w, use_pre_emphasis=True, use_delta_delta=True, use_pcen=True,
                                       use_spectral_contrast=True, n_contrast_bands=6)
    assert got.shape == ref.shape == (6, 110, 101)
    assert torch.isnan(ref[:, 103:]).all() and torch.isnan(got[:, 103:]).all()
    mel, rel = feature_errors(got[:, :103], ref[:, :103])
    assert mel < FEAT_TOL and rel < FEAT_TOL


def test_mel_only_configuration():
    p = cda.AudioPreprocessor(device="cuda", **{**SHIPPED, "use_mfcc": False})
    w = synth_batch(700, 8)
    got = p.extract_features(w.cuda()).cpu()
    ref = ofeat.extract_features_batch(w, use_mfcc=False)
    assert got.shape == ref.shape == (8, 64, 101)
    assert float((got - ref).abs().max()) < FEAT_TOL


@pytest.mark.parametrize("channels,n", [(1, 16000), (2, 20801), (3, 7999), (2, 16000), (1, 1), (2, 48000)])
def test_prepare_clip_is_bit_exact(pre, channels, n):
    """cough_prepare_clip = to_mono -> normalize -> pad_or_trim (preprocessing.py:185-212, :358-385): mean and
    division are single IEEE operations, so the result equals the restatement bit for bit."""
    g = torch.Generator().manual_seed(channels * 100003 + n)
    w = (torch.rand((channels, n), generator=g) - 0.5) * 1.7
    got = pre.prepare_clip(w.cuda()).cpu()
    want = ofeat.pad_or_trim(ofeat.normalize(ofeat.to_mono(w)))
    assert got.shape == want.shape == (1, 16000)
    assert torch.equal(got, want)
    raw = pre.prepare_clip(w.cuda(), normalize=False).cpu()
    assert torch.equal(raw, ofeat.pad_or_trim(ofeat.to_mono(w)))
    assert torch.equal(pre.prepare_clip(torch.zeros(channels, n).cuda()).cpu(), torch.zeros(1, 16000))   # no 0 / 0
    mono = pre.to_mono(w)                                      # the reference's helper methods run the same kernel
    assert not mono.is_cuda and torch.equal(mono, ofeat.to_mono(w))
    assert torch.equal(pre.normalize(mono), ofeat.normalize(ofeat.to_mono(w)))


def test_helper_methods_called_on_their_own():
    """apply_pre_emphasis / compute_deltas / apply_pcen are public methods of the reference class
    (/root/reference/src/preprocessing.py:214-240, :342-356, :305-340); extract_features fuses them, a caller may also use
    them directly.  Pre-emphasis and the deltas are single IEEE operations per element: bit-exact vs the restatement."""
    rng = np.random.default_rng(7)
    on = cda.AudioPreprocessor(device="cuda", **{**SHIPPED, "use_pre_emphasis": True})
    off = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    w = torch.from_numpy(rng.standard_normal((2, 16000)).astype(np.float32))
    assert off.apply_pre_emphasis(w) is w                                        # flag off: returned as is (:231-232)
    got = on.apply_pre_emphasis(w)
    assert not got.is_cuda and torch.equal(got, ofeat.pre_emphasis(w, 0.97))
    assert torch.equal(on.apply_pre_emphasis(w.cuda()).cpu(), got)
    x = torch.from_numpy(rng.standard_normal((3, 13, 101)).astype(np.float32))
    assert torch.equal(off.compute_deltas(x), ofeat.compute_deltas(x))
    assert torch.equal(off.compute_deltas(x[:, :, :1]), torch.zeros(3, 13, 1))   # one frame: (x - x) / 2
    mel = torch.from_numpy((rng.standard_normal((1, 64, 101)) ** 2 * 10).astype(np.float32))
    for kw in (dict(), dict(alpha=0.9, delta=1.5, r=0.4, eps=1e-5)):
        ref = ofeat.apply_pcen(mel, **kw)
        err = ((off.apply_pcen(mel, **kw) - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
        print(f"apply_pcen{kw}: rel err {err:.2e}")
        assert err < 1e-5
    with pytest.raises(ValueError, match="apply_pcen"):
        off.apply_pcen(mel[0])


@pytest.mark.parametrize("edges", [[3, 30, 130, 257], [10, 95, 257], [1, 3, 5, 11, 257], [0, 128, 257]],
                         ids=["27_and_100_bins", "85_bins", "narrow", "128_bins_from_dc"])
def test_contrast_selection_networks_of_every_size(monkeypatch, edges):
    """The sorted-slice sums of the contrast rows are register-resident selection networks instantiated for 1 .. 26 values per
    slice (spectrogram.hip: select_sums<CAP>); the reference's own logspace edges only reach a few of the sizes.  Same
    arithmetic with custom band edges (``contrast_edges`` is a field of the C-ABI's config), on both sides."""
    from cough_detector_amd import _tables
    n_bands = len(edges) - 2
    monkeypatch.setattr(_tables, "contrast_band_edges", lambda nb, nf: list(edges))
    monkeypatch.setattr(ofeat, "contrast_band_edges", lambda nb=6, nf=257: list(edges))
    flags = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=True,
                 n_contrast_bands=n_bands)
    p = cda.AudioPreprocessor(device="cuda", **flags)
    w = synth_batch(700, 12, peak_normalize=False)
    for normalize in (False, True):
        got = p.featurize_batch(w.cuda(), normalize=normalize).cpu()[:, 90:]
        ref = ofeat.extract_features_batch(w, normalize_first=normalize, **flags)[:, 90:]
        assert got.shape == ref.shape == (12, n_bands + 1, 101) and torch.isfinite(ref).all()
        err = (got - ref).abs().max().item()
        print(f"edges {edges} normalize={normalize}: abs err {err:.2e}")
        assert err < 2e-5


def test_mel_and_mfcc_helpers_transform_the_waveform_as_given():
    """extract_mel_spectrogram / extract_mfcc called on their own (/root/reference/src/preprocessing.py:387-430) do NOT apply
    pre-emphasis -- extract_features does that before calling them (:455-459) -- also on a preprocessor built with
    use_pre_emphasis=True; extract_spectral_contrast (:242-303) never sees the emphasised signal either."""
    flags = dict(use_pcen=False, use_pre_emphasis=True, use_delta_delta=True, use_spectral_contrast=True, n_contrast_bands=4)
    pre = cda.AudioPreprocessor(device="cuda", **flags)
    w = synth_batch(720, 1, peak_normalize=True)
    plain = ofeat.extract_features_batch(w, **{**flags, "use_pre_emphasis": False})
    emph = ofeat.extract_features_batch(w, **flags)
    assert (plain[:, :64] - emph[:, :64]).abs().max().item() > 1e-2             # the flag matters on this clip
    mel, mfcc, con = pre.extract_mel_spectrogram(w.cuda()), pre.extract_mfcc(w.cuda()), pre.extract_spectral_contrast(w.cuda())
    assert mel.shape == (1, 64, 101) and mfcc.shape == (1, 13, 101) and con.shape == (1, 5, 101)
    assert (mel.cpu() - plain[:, :64]).abs().max().item() < FEAT_TOL
    assert ((mfcc.cpu() - plain[:, 64:77]).abs() / plain[:, 64:77].abs().clamp(min=1.0)).max().item() < FEAT_TOL
    assert (con.cpu() - emph[:, -5:]).abs().max().item() < 2e-4                  # same rows with and without the flag
    full = pre.extract_features(w.cuda()).cpu()                                 # the whole image still is the emphasised one
    assert (full[:, :64] - emph[:, :64]).abs().max().item() < FEAT_TOL
    # PCEN branch of the helper (:400-404)
    pp = cda.AudioPreprocessor(device="cuda", **{**flags, "use_pcen": True, "use_spectral_contrast": False})
    ref = ofeat.extract_features_batch(w, use_pcen=True, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)
    assert (pp.extract_mel_spectrogram(w.cuda()).cpu() - ref[:, :64]).abs().max().item() < FEAT_TOL
