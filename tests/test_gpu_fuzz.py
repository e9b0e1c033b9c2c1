"""Randomised shapes through every C-ABI entry point that takes a batch: ragged batch sizes, strided inputs, other
image sizes.  Each case is checked against the CPU oracle (features / logits) or an exact host computation."""
import numpy as np
import pytest
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from oracle import cnn as ocnn, featurizer as ofeat, resnet as ores
from parity import FEAT_TOL, LOGIT_TOL, SHIPPED, feature_errors, realistic_state_dict, synth_batch

pytestmark = pytest.mark.gpu


def test_featuriser_random_batches_and_strides():
    rng = np.random.default_rng(77)
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    pool = synth_batch(1000, 40)
    ref_pool = ofeat.extract_features_batch(pool, normalize_first=True)
    for _ in range(12):
        b = int(rng.integers(1, 41))
        idx = torch.from_numpy(rng.permutation(40)[:b])
        stride = 16000 + 4 * int(rng.integers(0, 6))             # row stride must be a multiple of 4
        buf = torch.zeros((b, stride), device="cuda")
        buf[:, :16000] = pool[idx].cuda()
        got = pre.featurize_batch(buf[:, :16000], normalize=True).cpu()
        mel, rel = feature_errors(got, ref_pool[idx])
        assert mel < FEAT_TOL and rel < FEAT_TOL, (b, stride, mel, rel)


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16x3", LOGIT_TOL)])
def test_residual_net_random_image_sizes(dtype, tol):
    """The classifier is fully convolutional: other (F, T) sizes take the same f32 kernels with other shapes (the
    split-bf16 kernels are compiled for the shipped 90x101 image; a bf16x3 model runs other sizes on the exact-f32
    kernels).  Trained-scale head."""
    rng = np.random.default_rng(5)
    sd = realistic_state_dict(9)
    m = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype)
    m.load_state_dict(sd)
    m.cuda()
    import warnings
    for _ in range(6):
        b, f, t = int(rng.integers(1, 19)), int(rng.integers(24, 140)), int(rng.integers(24, 140))
        x = torch.rand((b, 1, f, t), generator=torch.Generator().manual_seed(b * 1000 + f))
        with warnings.catch_warnings(record=True) as rec:           # the fallback is announced, never silent
            warnings.simplefilter("always")
            got = m(x.cuda()).cpu()
        assert m.effective_dtype(f, t) == "fp32"
        assert (dtype == "bf16x3") == any("exact-f32 MFMA kernels instead" in str(w.message) for w in rec)
        want = ores.forward(x, sd)
        assert float((got - want).abs().max()) < tol, (b, f, t)
    with warnings.catch_warnings():                                   # the shipped image never warns
        warnings.simplefilter("error")
        m(torch.rand(2, 1, 90, 101).cuda())
    assert m.effective_dtype() == dtype


@pytest.mark.parametrize("kind", ["standard", "small"])
def test_conv_stack_random_image_sizes(cnn_golden, kind):
    rng = np.random.default_rng(6)
    sd, _ = cnn_golden[kind]
    m = cda.create_model(kind, n_mels=90, num_classes=2, in_channels=1)
    m.load_state_dict(sd)
    m.cuda()
    for _ in range(5):
        b, f, t = int(rng.integers(1, 14)), int(rng.integers(16, 120)), int(rng.integers(16, 120))
        x = torch.rand((b, 1, f, t), generator=torch.Generator().manual_seed(b * 77 + t))
        assert float((m(x.cuda()).cpu() - ocnn.FORWARD[kind](x, sd)).abs().max()) < 1e-4, (b, f, t)


def test_pipeline_every_small_batch_size():
    """Batches that are not multiples of the block kernels' clip groups (1, 2 or 3 clips per workgroup), of the STFT's
    8-clip slabs, or of anything else: fused pipeline == featurise -> classify, and both within tolerance of the
    oracle."""
    sd = realistic_state_dict(4)
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    wav = synth_batch(2000, 23)
    ref = ores.forward(ofeat.extract_features_batch(wav, normalize_first=True)[:, None], sd)
    for dtype, tol in (("bf16x3", LOGIT_TOL), ("bf16", 0.2), ("fp32", 1e-4)):   # "bf16": approximate mode, ~2 % of the spread
        m = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype)
        m.load_state_dict(sd)
        m.cuda()
        pipe = cda.CoughPipeline(pre, m)
        for b in list(range(1, 12)) + [17, 23]:
            got = pipe(wav[:b].cuda(), normalize=True)
            two_step = m(pre.featurize_batch(wav[:b].cuda(), normalize=True)[:, None])
            assert torch.equal(got, two_step), (dtype, b)
            assert float((got.cpu() - ref[:b]).abs().max()) < tol, (dtype, b)
            spec = pre.spectrogram_batch(wav[:b].cuda())
            assert spec.shape == (b, 257, 101) and bool(torch.isfinite(spec).all())


def test_stream_detector_reset_and_chunk_length_change():
    """The captured graphs are re-captured when the chunk length changes and survive reset(); results always equal
    the eager path's."""
    from cough_detector_amd.streaming import MultiStreamDetector
    sd = synth.random_state_dict(seed=8)
    streams = np.stack([synth.make_stream(60 + s, 5.0) for s in range(4)])
    out = []
    for use_graphs in (True, False):
        m = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16_approx")
        m.load_state_dict(sd)
        det = MultiStreamDetector(m, 4, confidence_threshold=2.0, clock=lambda: 0.0, use_graphs=use_graphs)
        log = []
        for chunk in (1600, 800, 4000):
            det.reset()
            for p in det.window_probs:
                p.clear()
            for i in range(0, 40000, chunk):
                det.push(streams[:, i:i + chunk])
            log.append([list(p) for p in det.window_probs])
        out.append(log)
    assert out[0] == out[1]
    assert all(len(p) == 7 for p in out[0][0])          # (40000 - 16000) / 4000 + 1 windows per stream


def test_featuriser_random_geometries_and_flags():
    """40 seeded random AudioPreprocessor constructor calls -- sample_rate, n_fft (powers of two), hop, win (odd / even / tiny),
    n_mels (more bands than bins included: empty bands), n_mfcc, f_min / f_max (up to Nyquist), window length, every flag --
    against the CPU oracle with the same arguments; the wide spectral-contrast bands of n_fft = 2048 included."""
    from test_oracle_featurizer import geometry_clip
    rng = np.random.default_rng(2026)
    for case in range(40):
        sr = int(rng.choice([8000, 11025, 16000, 22050, 32000, 44100]))
        n_fft = int(rng.choice([128, 256, 512, 512, 1024, 2048, 400, 300, 257, 1000]))
        win = int(rng.integers(max(8, n_fft // 8), n_fft + 1))
        hop = int(rng.integers(max(4, n_fft // 16), n_fft + 40))
        n_mels = int(rng.choice([13, 32, 40, 64, 80, 96, 128]))
        n_mfcc = int(rng.integers(1, min(n_mels, 40) + 1))
        f_min = float(rng.choice([0.0, 20.0, 100.0, 300.0]))
        f_max = float(min(sr / 2, rng.choice([3000.0, 4000.0, 8000.0, 16000.0])))
        if f_max <= f_min + 500:
            f_max = sr / 2
        n = int(rng.integers(n_fft // 2 + 1, 3 * sr // 2))
        flags = dict(use_pre_emphasis=bool(rng.integers(2)), use_delta_delta=bool(rng.integers(2)), use_pcen=bool(rng.integers(2)),
                     use_mfcc=bool(rng.integers(4) > 0), use_spectral_contrast=bool(rng.integers(3) == 0),
                     n_contrast_bands=int(rng.integers(1, 5)))
        g = dict(sample_rate=sr, n_mels=n_mels, n_fft=n_fft, hop_length=hop, win_length=win, f_min=f_min, f_max=f_max, n_mfcc=n_mfcc)
        pre = cda.AudioPreprocessor(device="cuda", segment_duration=n / sr, **g, **flags)
        n = pre.segment_samples
        w = torch.from_numpy(np.stack([geometry_clip(case + s, n) for s in (0, 3)]))
        f = pre.featurize_batch(w.cuda(), normalize=True).cpu()          # no combination in these ranges is refused
        ref = ofeat.extract_features_batch(w, normalize_first=True, **ofeat.geometry_kwargs(**g), **flags)
        assert f.shape == ref.shape, (case, g, flags, f.shape, ref.shape)
        nbase = n_mels + ((2 * n_mfcc + (n_mfcc if flags["use_delta_delta"] else 0)) if flags["use_mfcc"] else 0)
        mel = (f[:, :n_mels] - ref[:, :n_mels]).abs().max().item()
        zr = f[:, n_mels:nbase], ref[:, n_mels:nbase]
        rel = ((zr[0] - zr[1]).abs() / zr[1].abs().clamp(min=1.0)).max().item() if nbase > n_mels else 0.0
        cerr, nan_rule = 0.0, True
        if flags["use_spectral_contrast"]:
            fc, rc = f[:, nbase:], ref[:, nbase:]
            nan_rule = bool(torch.equal(torch.isnan(fc), torch.isnan(rc)))      # a one-bin first band: NaN rows in both (:272-300)
            both = ~torch.isnan(rc) & ~torch.isnan(fc)
            cerr = (fc[both] - rc[both]).abs().max().item() if both.any() else 0.0
        ok = mel < FEAT_TOL and rel < 3 * FEAT_TOL and cerr < 3 * FEAT_TOL and nan_rule and bool(torch.isfinite(f[:, :nbase]).all())
        print(f"case {case:2d}: sr {sr} n_fft {n_fft} win {win} hop {hop} mels {n_mels} mfcc {n_mfcc} f {f_min:.0f}-{f_max:.0f} N {n} "
              f"{[k[4:] for k, v in flags.items() if v is True]}: mel {mel:.1e} z {rel:.1e} contrast {cerr:.1e}{'' if ok else '   <-- FAIL'}")
        assert ok, (case, g, flags, mel, rel, cerr)


def test_realtime_preprocessor_random_chunk_lengths_and_hops():
    """RealtimePreprocessor.add_audio (src/preprocessing.py:582-616) with random chunk lengths (1 sample .. 1.7 windows: several
    windows per call, calls that complete none), hop durations and window lengths: the same windows in the same calls as the
    restated FIFO, features within tolerance; (1, n) and (n,) chunks; reset()."""
    rng = np.random.default_rng(99)
    for win_s, hop_s in ((1.0, 0.25), (1.0, 0.5), (0.5, 0.1), (2.0, 0.33)):
        rt = cda.RealtimePreprocessor(window_duration=win_s, hop_duration=hop_s, device="cuda", **SHIPPED)
        ow = ofeat.RealtimeWindowerOracle(win_s, hop_s)
        stream = torch.from_numpy(synth.make_stream(int(win_s * 10 + hop_s * 100), 4.0 * win_s + 1.0))
        pos, n_win, calls_with_many = 0, 0, 0
        while pos < stream.numel():
            n = int(rng.choice([1, 7, 160, 1600, 4000, int(16000 * win_s * 1.7)]))
            chunk = stream[pos:pos + n]
            pos += n
            if rng.integers(2):
                chunk = chunk.unsqueeze(0)                                   # (1, n) as well as (n,)
            got, want = rt.add_audio(chunk), ow.add_audio(chunk)
            assert len(got) == len(want), (win_s, hop_s, pos, len(got), len(want))
            calls_with_many += len(got) > 1
            for g, r in zip(got, want):
                mel, rel = feature_errors(g, r)
                assert g.shape == r.shape and mel < FEAT_TOL and rel < FEAT_TOL
                n_win += 1
        assert n_win >= 8 and calls_with_many >= 1 and rt.buffer.shape == ow.buffer.shape
        rt.reset()
        assert rt.buffer.shape == (1, 0) and rt.add_audio(stream[:100]) == []


def test_tuned_kernel_every_flag_combination_on_hard_inputs():
    """The shipped geometry (tuned kernels: K1 with its PCEN / pre-emphasis / delta-delta branches, K7 x 2 + K8 for the contrast
    rows) under all 32 combinations of the constructor's boolean flags, 3 contrast bands, on inputs chosen to hit the corners:
    digital silence, DC, a unit impulse, a full-scale square wave, 1e-6-amplitude noise, a synthetic cough, a stereo-like
    alternating pattern, a clip that is silent except for its last sample."""
    rng = np.random.default_rng(31)
    n = 16000
    hard = np.zeros((8, n), dtype=np.float32)
    hard[1] = 0.25                                                          # DC
    hard[2, 8000] = 1.0                                                     # impulse
    hard[3] = np.where((np.arange(n) // 40) % 2 == 0, 1.0, -1.0)            # full-scale square wave, 200 Hz
    hard[4] = (rng.standard_normal(n) * 1e-6).astype(np.float32)           # far below amin once squared, before normalise
    hard[5] = synth.make_clip(12345, peak_normalize=False) * 0.3
    hard[6] = np.where(np.arange(n) % 2 == 0, 0.5, -0.5)                    # Nyquist tone
    hard[7, -1] = -0.7
    w = torch.from_numpy(hard)
    import itertools
    import warnings
    worst = worst_ill = 0.0
    for pe, dd, pc, mf, sc in itertools.product([False, True], repeat=5):
        kw = dict(use_pre_emphasis=pe, use_delta_delta=dd, use_pcen=pc, use_mfcc=mf, use_spectral_contrast=sc, n_contrast_bands=3)
        pre = cda.AudioPreprocessor(device="cuda", **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            f = pre.featurize_batch(w.cuda(), normalize=True).cpu()
            ref = ofeat.extract_features_batch(w, normalize_first=True, **kw)
        assert f.shape == ref.shape, (kw, f.shape, ref.shape)
        nan_f, nan_r = torch.isnan(f), torch.isnan(ref)
        assert torch.equal(nan_f, nan_r), (kw, int(nan_f.sum()), int(nan_r.sum()))   # e.g. 0 / 0 rows of a silent clip, as the reference
        d = torch.where(nan_r, torch.zeros_like(f), (f - ref).abs() / ref.abs().clamp(min=1.0)).flatten(1).amax(dim=1)
        # DC (1) and the Nyquist tone (6) put NO energy into the 100-4000 Hz filterbank: every mel value is float32 rounding
        # noise of the leakage, the per-clip max that anchors the top_db floor included -- the f32 CPU oracle itself is
        # 7.7e-4 / 1.4e-3 away from the float64 re-derivation on them.  Checked for shape / NaN rule / boundedness only.
        ill = torch.tensor([False, True, False, False, False, False, True, False])
        worst_ill = max(worst_ill, d[ill].max().item())
        assert d[ill].max().item() < 0.1, (kw, d)                 # measured: up to 1.1e-2
        err = d[~ill].max().item()
        worst = max(worst, err)
        assert err < 3 * FEAT_TOL, (kw, d)
    print(f"32 flag combinations x 8 hard inputs at the shipped geometry: worst error on the 6 well-conditioned inputs {worst:.2e} "
          f"(the 2 ill-conditioned ones: {worst_ill:.2e})")


def test_fullband_featuriser_random_filterbanks_and_flags():
    """40 seeded random filterbanks at the shipped STFT geometry -- n_mels (even, 2..128), n_mfcc (1..20), f_min / f_max up to the
    Nyquist bin, every flag the full-band one-launch kernel takes (PCEN at 64 bands) -- against the CPU oracle; every case must
    land on the one-launch kernel (`tuned` or `tuned_fullband`), none on the generic chain."""
    rng = np.random.default_rng(int(__import__("os").environ.get("COUGH_FUZZ_SEED_FULLBAND", "505")))
    w = synth_batch(2100, 10, peak_normalize=False)
    paths = {"tuned": 0, "tuned_fullband": 0}
    for case in range(40):
        n_mels = int(rng.choice([2, 8, 20, 26, 40, 64, 64, 64, 80, 96, 128]))
        n_mfcc = int(rng.integers(1, min(n_mels, 20) + 1))
        f_min = float(rng.choice([0.0, 20.0, 100.0, 300.0, 1000.0]))
        f_max = float(rng.choice([2000.0, 3500.0, 4000.0, 5000.0, 7000.0, 8000.0]))
        flags = dict(use_pre_emphasis=bool(rng.integers(2)), use_delta_delta=bool(rng.integers(2)),
                     use_pcen=bool(rng.integers(2)) and n_mels == 64, use_mfcc=bool(rng.integers(4) > 0),
                     use_spectral_contrast=bool(rng.integers(4) == 0), n_contrast_bands=int(rng.integers(1, 5)))
        g = dict(sample_rate=16000, n_mels=n_mels, n_fft=512, hop_length=160, win_length=400, f_min=f_min, f_max=f_max, n_mfcc=n_mfcc)
        pre = cda.AudioPreprocessor(device="cuda", **g, **flags)
        assert pre.kernel_path() in paths, (case, g, flags, pre.kernel_path())
        paths[pre.kernel_path()] += 1
        normalize = bool(rng.integers(2))
        f = pre.featurize_batch(w.cuda(), normalize=normalize).cpu()
        ref = ofeat.extract_features_batch(w, normalize_first=normalize, **ofeat.geometry_kwargs(**g), **flags)
        assert f.shape == ref.shape, (case, g, flags)
        nbase = n_mels + ((2 * n_mfcc + (n_mfcc if flags["use_delta_delta"] else 0)) if flags["use_mfcc"] else 0)
        mel = (f[:, :n_mels] - ref[:, :n_mels]).abs().max().item()
        zr = f[:, n_mels:nbase], ref[:, n_mels:nbase]
        rel = ((zr[0] - zr[1]).abs() / zr[1].abs().clamp(min=1.0)).max().item() if nbase > n_mels else 0.0
        cerr = (f[:, nbase:] - ref[:, nbase:]).abs().max().item() if flags["use_spectral_contrast"] else 0.0
        assert mel < 1e-4 and rel < 1e-4 and cerr < 2e-4, (case, g, flags, normalize, mel, rel, cerr)
    print(f"full-band fuzz: {paths}")
    assert paths["tuned_fullband"] >= 30


def test_runtime_geometry_featuriser_random_stft_geometries():
    """50 seeded random STFT geometries at n_fft = 512 -- sample rate, hop (1..256), window (1..512), segment length (so that the
    frame count stays <= 128), filterbank and flags -- against the CPU oracle; each must land on the one-launch kernel with the
    run-time geometry unless the case hits one of its stated limits (MFCC rows beyond 16 640 B; the dB buffer's LDS),
    which the test computes itself."""
    from test_oracle_featurizer import geometry_clip
    rng = np.random.default_rng(int(__import__("os").environ.get("COUGH_FUZZ_SEED_GEO", "606")))
    paths = {"tuned_geometry": 0, "generic": 0, "tuned_fullband": 0, "tuned": 0}
    for case in range(50):
        sr = int(rng.choice([8000, 11025, 16000, 22050, 32000, 44100, 48000]))
        hop = int(rng.choice([rng.integers(3, 257), 64, 128, 160, 200, 220, 256]))
        win = int(rng.choice([rng.integers(1, 513), 256, 400, 441, 512]))
        T = int(rng.integers(max(2, 257 // hop + 2), 129 if case % 3 else 260))
        n = (T - 1) * hop + int(rng.integers(0, hop))
        if n <= 256:
            continue
        n_mels = int(rng.choice([2, 8, 20, 40, 64, 64, 64, 80, 96]))
        n_mfcc = int(rng.integers(1, min(n_mels, 20) + 1))
        f_min = float(rng.choice([0.0, 20.0, 100.0, 300.0]))
        f_max = float(rng.choice([0.25, 0.4, 0.5]) * sr)
        flags = dict(use_pre_emphasis=bool(rng.integers(2)), use_delta_delta=bool(rng.integers(2)),
                     use_pcen=bool(rng.integers(3) == 0), use_mfcc=bool(rng.integers(4) > 0),
                     use_spectral_contrast=bool(rng.integers(4) == 0), n_contrast_bands=int(rng.integers(1, 5)))
        g = dict(sample_rate=sr, n_mels=n_mels, n_fft=512, hop_length=hop, win_length=win, f_min=f_min, f_max=f_max, n_mfcc=n_mfcc)
        pre = cda.AudioPreprocessor(device="cuda", segment_duration=(n + 0.5) / sr, **g, **flags)
        assert pre.segment_samples == n and pre._frames(n) == T, (case, g, n, T)
        shipped = (sr, hop, win, n) == (16000, 160, 400, 16000)
        limits = (flags["use_mfcc"] and n_mfcc * T * 4 > 16640) or (flags["use_pcen"] and T > 208) or \
            n_mels * T * 4 > 48 * 1024
        if not shipped and not limits:
            assert pre.kernel_path() == "tuned_geometry", (case, g, flags, n, T)
        paths[pre.kernel_path()] += 1
        w = torch.from_numpy(np.stack([geometry_clip(case + s, n) for s in range(4)]))
        normalize = bool(rng.integers(2))
        f = pre.featurize_batch(w.cuda(), normalize=normalize).cpu()
        ref = ofeat.extract_features_batch(w, normalize_first=normalize, **ofeat.geometry_kwargs(**g), **flags)
        assert f.shape == ref.shape == (4, pre.get_num_features(), T), (case, g, flags)
        nbase = f.shape[1] - (flags["n_contrast_bands"] + 1 if flags["use_spectral_contrast"] else 0)
        mel = (f[:, :n_mels] - ref[:, :n_mels]).abs().max().item()
        rel = ((f[:, n_mels:nbase] - ref[:, n_mels:nbase]).abs() / ref[:, n_mels:nbase].abs().clamp(min=1.0)).max().item() if nbase > n_mels else 0.0
        cerr = (f[:, nbase:] - ref[:, nbase:]).abs().max().item() if flags["use_spectral_contrast"] else 0.0
        assert mel < 1e-4 and rel < 2e-4 and cerr < 2e-4, (case, g, flags, n, T, normalize, pre.kernel_path(), mel, rel, cerr)
    print(f"run-time geometry fuzz: {paths}")
    assert paths["tuned_geometry"] >= 30


def test_random_waveform_lengths_through_random_handles():
    """extract_features of any length (/root/reference/src/preprocessing.py:432-489 never checks it) through handles of random
    n_fft-512 configurations: lengths that fit the run-time-geometry kernel's limits take it, the others the generic chain --
    every result against the CPU oracle, and the handle's own segment is unharmed in between."""
    from test_oracle_featurizer import geometry_clip
    import os
    rng = np.random.default_rng(int(os.environ.get("COUGH_FUZZ_SEED", "707")))      # tools/soak_any_length.sh: more seeds / cases
    for case in range(int(os.environ.get("COUGH_FUZZ_CASES", "12"))):
        n_mels = int(rng.choice([20, 40, 63, 64, 64, 64, 80, 128]))
        n_mfcc = int(rng.integers(1, min(n_mels, 30) + 1))
        hop = int(rng.choice([100, 128, 160, 160, 160, 200, 256, 300]))
        win = int(rng.choice([256, 400, 400, 512]))
        f_max = float(rng.choice([3000.0, 4000.0, 8000.0]))
        flags = dict(use_pre_emphasis=bool(rng.integers(2)), use_delta_delta=bool(rng.integers(2)), use_pcen=bool(rng.integers(3) == 0),
                     use_mfcc=bool(rng.integers(4) > 0), use_spectral_contrast=bool(rng.integers(4) == 0),
                     n_contrast_bands=int(rng.integers(1, 5)))
        g = dict(sample_rate=16000, n_mels=n_mels, n_fft=512, hop_length=hop, win_length=win, f_min=100.0, f_max=f_max, n_mfcc=n_mfcc)
        pre = cda.AudioPreprocessor(device="cuda", **g, **flags)
        seg = torch.from_numpy(np.stack([geometry_clip(case, 16000)]))
        base = pre.featurize_batch(seg.cuda(), normalize=True)
        kw = ofeat.geometry_kwargs(**g)
        for n in [int(v) for v in rng.integers(257, 45000, size=3)]:
            w = torch.from_numpy(np.stack([geometry_clip(case + s, n) for s in range(3)]))
            normalize = bool(rng.integers(2))
            f = pre.featurize_batch(w.cuda(), normalize=normalize).cpu()
            ref = ofeat.extract_features_batch(w, normalize_first=normalize, **kw, **flags)
            assert f.shape == ref.shape == (3, pre.get_num_features(), n // hop + 1), (case, g, flags, n)
            nbase = f.shape[1] - (flags["n_contrast_bands"] + 1 if flags["use_spectral_contrast"] else 0)
            mel = (f[:, :n_mels] - ref[:, :n_mels]).abs().max().item()
            rel = ((f[:, n_mels:nbase] - ref[:, n_mels:nbase]).abs() / ref[:, n_mels:nbase].abs().clamp(min=1.0)).max().item() if nbase > n_mels else 0.0
            cerr = (f[:, nbase:] - ref[:, nbase:]).abs().max().item() if flags["use_spectral_contrast"] else 0.0
            assert mel < 1e-4 and rel < 2e-4 and cerr < 2e-4, (case, g, flags, n, normalize, pre.kernel_path(), mel, rel, cerr)
        assert torch.equal(pre.featurize_batch(seg.cuda(), normalize=True), base), (case, g, flags)
