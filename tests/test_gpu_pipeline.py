"""Fused pipeline (cough_pipeline_forward) vs the two-step path and the CPU oracle."""
import pytest
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from oracle import featurizer as ofeat, resnet as ores
from parity import FEAT_TOL, LOGIT_TOL, SHIPPED, feature_errors, synth_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16", "fp32"])
@pytest.mark.parametrize("normalize", [True, False])
def test_pipeline_equals_two_step(resnet_golden, dtype, normalize):
    sd, _ = resnet_golden
    w = synth_batch(700, 37, peak_normalize=False).cuda() * 0.4          # 37: ragged last clip groups (G = 2, 3)
    w[3] = 0.0
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype)
    model.load_state_dict(sd)
    model.cuda()
    pipe = cda.CoughPipeline(pre, model)
    feats = pre.featurize_batch(w, normalize=normalize)
    want = model(feats.unsqueeze(1))
    got = pipe(w, normalize=normalize)
    assert torch.equal(got, want)                                        # same kernels' arithmetic, fused or not
    got2, f2 = pipe(w, normalize=normalize, return_features=True)
    assert torch.equal(got2, want) and torch.equal(f2, feats)
    preds, probs = pipe.predict(w, normalize=normalize)
    assert torch.equal(preds, want.argmax(1))
    assert (probs - torch.softmax(want, 1)).abs().max() < 1e-6
    assert pipe(w[:0]).shape == (0, 2)


def test_pipeline_full_batch_against_oracle(resnet_golden):
    sd, _ = resnet_golden
    B = 4096
    w = synth_batch(9000, B, peak_normalize=False)
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(sd)
    pipe = cda.CoughPipeline(pre, model.cuda())
    logits = pipe(w.cuda(), normalize=True)
    assert torch.equal(pipe(w.cuda()[:64], normalize=True), logits[:64])  # batch invariance
    sample = torch.arange(0, B, 32)
    ref = ores.forward(ofeat.extract_features_batch(w[sample], normalize_first=True).unsqueeze(1), sd)
    err = (logits[sample].cpu() - ref).abs().max().item()
    print(f"pipeline bf16x3 B=4096 sample: logits max abs err {err:.2e}")
    assert err < LOGIT_TOL
    flips = int((logits[sample].cpu().argmax(1) != ref.argmax(1)).sum())
    print(f"pipeline bf16x3 B=4096: argmax differs on {flips} of {len(sample)} sampled clips (no mask; smallest reference margin "
          f"{(ref[:, 1] - ref[:, 0]).abs().min().item():.2e})")
    assert flips == 0                                                     # argmax exact on EVERY sampled clip, no margin mask
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record(); ev[1].record()
    pipe(w.cuda(), events=ev)
    torch.cuda.synchronize()
    assert 0.0 < ev[0].elapsed_time(ev[1]) < 50.0


@pytest.mark.parametrize("flags,rows", [(dict(use_delta_delta=True), 103), (dict(use_mfcc=False), 64),
                                        (dict(use_spectral_contrast=True, n_contrast_bands=4), 95),
                                        (dict(use_delta_delta=True, use_pre_emphasis=True, use_pcen=True,
                                              use_spectral_contrast=True, n_contrast_bands=4), 108)])
def test_pipeline_with_the_reference_default_flags_runs_the_split_bf16_blocks(resnet_heights_golden, flags, rows):
    """waveform -> logits in one C-ABI call when the stem cannot be fused (delta-delta / PCEN / contrast rows): features are
    materialised in the workspace, the 103- and 108-row images run on the split-bf16 kernels (block inputs 26x25 / 27x25),
    and the logits match featurise + CPU oracle classifier."""
    import warnings
    sd, _ = resnet_heights_golden["h103"]
    w = synth_batch(900, 21, peak_normalize=False) * 0.5
    kw = {"use_mfcc": True, **SHIPPED, **flags}
    pre = cda.AudioPreprocessor(device="cuda", **kw)
    model = cda.create_model("residual", n_mels=rows, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(sd)
    pipe = cda.CoughPipeline(pre, model.cuda())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        logits, feats = pipe(w.cuda(), normalize=True, return_features=True)
    assert feats.shape == (21, rows, 101) and model.effective_dtype(rows, 101) == "bf16x3"
    ref_feats = ofeat.extract_features_batch(w, normalize_first=True, **kw)
    ref = ores.forward(ref_feats.unsqueeze(1), sd)
    err = (logits.cpu() - ref).abs().max().item()
    print(f"{rows}-row pipeline: logits max abs err {err:.2e}")
    assert err < LOGIT_TOL
    assert torch.equal(pipe(w.cuda(), normalize=True), logits)


def test_pipeline_batch_beyond_2_gib_of_activations(resnet_golden):
    """40 000 clips in ONE call: the waveforms (2.56 GB), the stem output (2.8 GB) and the feature image (1.45 GB) each pass
    2^31 bytes, so every clip offset in the kernels has to be 64-bit.  Size-independent property: batch invariance -- the last
    512 clips (the ones past the 2 GiB marks) equal a 512-clip call bit for bit, and a sample matches the CPU oracle."""
    sd, _ = resnet_golden
    B = 40000
    w = synth.device_clips(7_000_000, B)
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(sd)
    pipe = cda.CoughPipeline(pre, model.cuda())
    logits, feats = pipe(w, normalize=True, return_features=True)
    tail_logits, tail_feats = pipe(w[-512:], normalize=True, return_features=True)
    assert torch.equal(logits[-512:], tail_logits) and torch.equal(feats[-512:], tail_feats)
    assert torch.equal(pipe(w, normalize=True), logits)                   # features not materialised: same logits
    two_step = model(pre.featurize_batch(w, normalize=True).unsqueeze(1))
    assert torch.equal(two_step, logits)
    idx = torch.tensor([0, 1, 16383, 16384, 32767, 32768, 33554, 39998, 39999])
    ref = ores.forward(ofeat.extract_features_batch(w[idx].cpu(), normalize_first=True).unsqueeze(1), sd)
    assert (logits[idx].cpu() - ref).abs().max().item() < LOGIT_TOL


def test_reference_default_constructor_gives_nan_logits_as_the_reference_does(resnet_heights_golden):
    """AudioPreprocessor() with every default (6 contrast bands): the reference's contrast rows are NaN by construction
    (/root/reference/src/preprocessing.py:272-300), its ReLU / max-pool propagate the NaN, and every logit is NaN -- its engine
    never fires in that configuration.  Same here, on the pipeline and on model(features)."""
    import warnings
    sd, _ = resnet_heights_golden["h110"]
    w = synth_batch(300, 6, peak_normalize=False) * 0.5
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pre = cda.AudioPreprocessor(device="cuda")
        model = cda.create_model("residual", n_mels=110, num_classes=2, in_channels=1, compute_dtype="bf16x3")
        model.load_state_dict(sd)
        pipe = cda.CoughPipeline(pre, model.cuda())
        logits, feats = pipe(w.cuda(), normalize=True, return_features=True)
        ref = ores.forward(ofeat.extract_features_batch(w, normalize_first=True, use_pre_emphasis=True, use_delta_delta=True,
                                                        use_pcen=True, use_spectral_contrast=True, n_contrast_bands=6).unsqueeze(1), sd)
        preds, probs = pipe.predict(w.cuda(), normalize=True)
    assert feats.shape == (6, 110, 101) and torch.isnan(feats[:, 103:]).all() and torch.isfinite(feats[:, :103]).all()
    assert torch.isnan(ref).all() and torch.isnan(logits).all() and torch.isnan(model(feats.unsqueeze(1))).all()
    assert torch.isnan(probs).all() and int(preds.abs().sum()) == 0


@pytest.mark.parametrize("normalize", [False, True])
def test_delta_delta_pipeline_keeps_the_stem_fused_in_two_halves(resnet_heights_golden, normalize):
    """the 103-row image of the reference's delta-delta flag (src/preprocessing.py:43-49, :471-474) no longer
    leaves the CU -- the split-bf16 stem runs inside the featurise kernel, in two halves of 13 pooled rows.  Logits are
    bit-identical to featurise -> classify (same images, same MFMA sequence per output), within 1e-3 of the CPU oracle, the
    optional feature output is still the full image, and a NaN sample still gives NaN logits for that clip only."""
    sd, _ = resnet_heights_golden["h103"]
    kw = {**SHIPPED, "use_delta_delta": True}
    pre = cda.AudioPreprocessor(device="cuda", **kw)
    model = cda.create_model("residual", n_mels=103, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(sd)
    model.cuda().eval()
    pipe = cda.CoughPipeline(pre, model)
    w = synth_batch(1200, 70, peak_normalize=False) * 0.7
    logits, feats = pipe(w.cuda(), normalize=normalize, return_features=True)
    direct = pre.featurize_batch(w.cuda(), normalize=normalize)
    assert torch.equal(feats, direct)
    assert torch.equal(logits, model(direct[:, None]))                     # bit-identical to the two-step path
    assert torch.equal(pipe(w.cuda(), normalize=normalize), logits)       # ... and without materialised features
    ref = ores.forward(ofeat.extract_features_batch(w, normalize_first=normalize, **kw).unsqueeze(1), sd)
    err = (logits.cpu() - ref).abs().max().item()
    print(f"103-row fused pipeline normalize={normalize}: logits max abs err {err:.2e}")
    assert err < LOGIT_TOL and torch.equal(logits.cpu().argmax(1), ref.argmax(1))
    bad = w.clone()
    bad[3, 5000] = float("nan")
    lb = pipe(bad.cuda(), normalize=normalize).cpu()
    assert torch.isnan(lb[3]).all() and torch.equal(lb[[0, 1, 2, 4, 69]], logits.cpu()[[0, 1, 2, 4, 69]])


@pytest.mark.parametrize("kw", [dict(use_pre_emphasis=True), dict(use_pre_emphasis=True, use_delta_delta=True),
                                dict(f_max=8000.0, use_delta_delta=True), dict(f_max=8000.0, use_pre_emphasis=True),
                                dict(f_max=8000.0, use_pre_emphasis=True, use_delta_delta=True),
                                dict(use_pcen=True), dict(use_pcen=True, use_pre_emphasis=True, use_delta_delta=True),
                                dict(f_max=8000.0, use_pcen=True, use_delta_delta=True)],
                         ids=["preemph", "preemph_dd", "fullband_dd", "fullband_preemph", "fullband_preemph_dd", "pcen",
                              "pcen_preemph_dd", "fullband_pcen_dd"])
def test_every_fused_stem_instantiation_equals_featurise_then_classify(resnet_golden, resnet_heights_golden, kw):
    """The split-bf16 stem stays inside the featurise kernel for pre-emphasis, PCEN mel rows, any filterbank at the shipped STFT geometry and
    for the 103-row delta-delta image, in every combination (featurize_kernel<PRE_EMPH, 2, FULL, TALL>): bit-identical to the
    two-step path, within 1e-3 of the CPU oracle."""
    flags = {**SHIPPED, **{k: v for k, v in kw.items() if k.startswith("use_")}}
    geom = {k: v for k, v in kw.items() if not k.startswith("use_")}
    rows = 103 if flags["use_delta_delta"] else 90
    sd = resnet_heights_golden["h103"][0] if rows == 103 else resnet_golden[0]
    pre = cda.AudioPreprocessor(device="cuda", **geom, **flags)
    model = cda.create_model("residual", n_mels=rows, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(sd)
    model.cuda().eval()
    pipe = cda.CoughPipeline(pre, model)
    w = synth_batch(1300, 45, peak_normalize=False) * 0.6
    logits = pipe(w.cuda(), normalize=True)
    feats = pre.featurize_batch(w.cuda(), normalize=True)
    assert feats.shape == (45, rows, 101) and torch.equal(logits, model(feats[:, None]))
    g = dict(sample_rate=16000, n_mels=64, n_fft=512, hop_length=160, win_length=400, f_min=100.0, f_max=4000.0, n_mfcc=13)
    g.update(geom)
    ref = ores.forward(ofeat.extract_features_batch(w, normalize_first=True, **flags, **ofeat.geometry_kwargs(**g)).unsqueeze(1), sd)
    err = (logits.cpu() - ref).abs().max().item()
    print(f"{kw}: fused pipeline logits max abs err {err:.2e}")
    assert err < LOGIT_TOL and torch.equal(logits.cpu().argmax(1), ref.argmax(1))
