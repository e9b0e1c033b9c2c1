"""CoughDetector ("standard") and CoughDetectorSmall on the HIP path (csrc/cnn.hip, through the C-ABI) against goldens
produced by the REFERENCE modules (oracle/make_golden_cnn.py) and against the CPU oracle on fresh inputs."""
import pytest
import torch

import cough_detector_amd as cda
from oracle import cnn as ocnn
from parity import LOGIT_TOL, synth_batch

pytestmark = pytest.mark.gpu

KINDS = ["standard", "small"]


def _model(kind, sd, dtype):
    m = cda.create_model(kind, n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype)
    m.load_state_dict(sd)
    return m.cuda().eval()


@pytest.mark.parametrize("kind", KINDS)
def test_fp32_matches_reference_goldens(cnn_golden, kind):
    sd, vec = cnn_golden[kind]
    m = _model(kind, sd, "fp32")
    x = cnn_golden["x"].cuda()
    conv = m.conv_output(x).cpu()
    assert conv.shape == vec["conv_out"].shape
    cerr = float((conv - vec["conv_out"]).abs().max())
    logits = m(x).cpu()
    lerr = float((logits - vec["logits"]).abs().max())
    preds, probs = m.predict(x)
    print(f"{kind} fp32: conv_out {cerr:.2e}, logits {lerr:.2e}")
    assert cerr < 1e-4 and lerr < LOGIT_TOL
    assert torch.equal(preds.cpu(), vec["preds"]) and preds.dtype == torch.int64
    assert float((probs.cpu() - vec["probs"]).abs().max()) < 1e-4      # trained-scale head: logits agree to ~1e-4


@pytest.mark.parametrize("kind", KINDS)
def test_bf16x3_matches_reference_goldens_at_the_logit_tolerance(cnn_golden, kind):
    """The split-bf16 path of the conv-stack nets (cnn_conv_lds_x3_kernel: hi + lo operands, three MFMAs per k-step, f32
    activations in HBM) is parity-grade: logits within 1e-3 of the REFERENCE goldens at a trained head's scale, argmax
    equal on every clip, no margin mask, no relative tolerance."""
    sd, vec = cnn_golden[kind]
    m = _model(kind, sd, "bf16x3")
    x = cnn_golden["x"].cuda()
    conv = m.conv_output(x).cpu()
    cerr = float((conv - vec["conv_out"]).abs().max() / vec["conv_out"].abs().max())
    logits = m(x).cpu()
    lerr = float((logits - vec["logits"]).abs().max())
    preds, probs = m.predict(x)
    print(f"{kind} bf16x3: conv_out {cerr:.2e} (relative to max), logits {lerr:.2e}")
    assert cerr < 1e-4 and lerr < LOGIT_TOL
    assert torch.equal(preds.cpu(), vec["preds"])
    assert float((probs.cpu() - vec["probs"]).abs().max()) < 1e-3


# compute_dtype="bf16_approx" of the conv-stack nets is a fast APPROXIMATE mode (single bf16 operands, bf16 activations): on a
# trained-scale head its logit error is a few percent of the class-margin spread, outside LOGIT_TOL.  The parity-grade
# mode of these two classifiers is "fp32" (tests above / below at 1e-4 .. 1e-3).
BF16_REL = 0.12        # approximate mode: bound on max |class-margin error| / margin spread (and on |logit error| / max |logit|)


def _approx_ok(got, ref):
    gm, rm = got[:, 1] - got[:, 0], ref[:, 1] - ref[:, 0]
    return (float((gm - rm).abs().max()) < BF16_REL * float(rm.std()) and
            float((got - ref).abs().max()) < BF16_REL * float(ref.abs().max()))


@pytest.mark.parametrize("kind", KINDS)
def test_bf16_is_an_approximate_mode(cnn_golden, kind):
    sd, vec = cnn_golden[kind]
    m = _model(kind, sd, "bf16_approx")
    x = cnn_golden["x"].cuda()
    logits = m(x).cpu()
    lerr = float((logits - vec["logits"]).abs().max())
    spread = float((vec["logits"][:, 1] - vec["logits"][:, 0]).std())
    print(f"{kind} bf16: logits {lerr:.2e} = {100 * lerr / spread:.1f} % of the margin spread {spread:.2f}")
    assert _approx_ok(logits, vec["logits"])
    margin = (vec["logits"][:, 1] - vec["logits"][:, 0]).abs()
    safe = margin > 2 * lerr
    preds, _ = m.predict(x)
    assert torch.equal(preds.cpu()[safe], vec["preds"][safe])


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("dtype,tol", [("fp32", LOGIT_TOL), ("bf16x3", LOGIT_TOL), ("bf16_approx", None)])
def test_fresh_features_ragged_batch_and_other_sizes(cnn_golden, kind, dtype, tol):
    """A batch that is not a multiple of any tile (37 clips), real featuriser output, and a second image size
    (the networks are fully convolutional: global mean at the end)."""
    sd, _ = cnn_golden[kind]
    m = _model(kind, sd, dtype)
    pre = cda.AudioPreprocessor(device="cuda", use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                                use_spectral_contrast=False)
    feats = pre.featurize_batch(synth_batch(900, 37).cuda())[:, None]            # (37, 1, 90, 101)
    ref = ocnn.FORWARD[kind](feats.cpu(), sd)
    got = m(feats).cpu()
    assert got.shape == (37, 2)
    small_img = feats[:5, :, :64, :47].contiguous()
    got_small, ref_small = m(small_img).cpu(), ocnn.FORWARD[kind](small_img.cpu(), sd)
    if tol is None:                                       # approximate mode: relative to the spread of these logits
        assert _approx_ok(got, ref)
        assert float((got_small - ref_small).abs().max()) < BF16_REL * float(ref.abs().max())
    else:
        assert float((got - ref).abs().max()) < tol
        assert float((got_small - ref_small).abs().max()) < tol
    assert m(feats[:0]).shape == (0, 2)
    # batch invariance: per-clip results do not depend on the neighbours
    assert torch.equal(m(feats[3:4]).cpu(), got[3:4])


def test_interface_errors(cnn_golden):
    sd, _ = cnn_golden["small"]
    with pytest.warns(UserWarning, match="APPROXIMATE single-bf16 mode"):      # nobody gets it without being told
        assert cda.create_model("small", compute_dtype="bf16").compute_dtype == "bf16_approx"
    with pytest.raises(ValueError, match="compute_dtype must be one of 'fp32', 'bf16x3', 'bf16_approx'"):
        cda.create_model("standard", compute_dtype="fp16")
    m = _model("small", sd, "fp32")
    with pytest.raises(ValueError, match="expected input"):
        m(torch.zeros(2, 90, 101))
    with pytest.raises(ValueError, match="too small"):
        m(torch.zeros(1, 1, 4, 4))
    with pytest.raises(RuntimeError, match="inference-only"):
        m.train()(torch.zeros(1, 1, 90, 101))
    with pytest.raises(ValueError, match="Unknown model type"):
        cda.create_model("huge")


def test_narrow_channel_tuples_load_under_bf16x3_like_under_fp32():
    """CoughDetector(channels=...) with an 8- or 24-wide first block loaded under fp32 and must not fail under the
    engine's default bf16x3: layers the split-bf16 kernel has no instantiation for run on the exact-f32 kernel (f32
    activations either way).  /root/reference/src/model.py:50-92 takes any channels tuple."""
    torch.manual_seed(5)
    x = torch.from_numpy(__import__("numpy").load(
        __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "features_golden.npz"))["features"][:6])[:, None]
    for channels in ((8, 32, 64), (24, 32, 64, 128), (16, 32, 64)):   # later blocks: cout in {32, 64, 128 k} in every mode
        ref_model = None
        sd = None
        for dtype in ("fp32", "bf16x3"):
            m = cda.CoughDetector(n_mels=90, channels=channels, compute_dtype=dtype)
            if sd is None:
                with torch.no_grad():                                   # non-trivial BatchNorm statistics
                    for name, buf in m.named_buffers():
                        if name.endswith("running_mean"):
                            buf.normal_(0.0, 0.3)
                        elif name.endswith("running_var"):
                            buf.uniform_(0.5, 1.5)
                sd = {k: v.clone() for k, v in m.state_dict().items()}
                ref_model = ocnn.standard_forward(x, sd)
            m.load_state_dict(sd)
            got = m.cuda().eval()(x.cuda()).cpu()
            err = float((got - ref_model).abs().max())
            print(f"channels {channels} {dtype}: logits max abs err {err:.2e}")
            assert err < LOGIT_TOL


@pytest.mark.parametrize("kind,dtype", [("standard", "bf16x3"), ("small", "fp32")])
def test_conv_stack_batch_beyond_2_gib_of_activations(kind, dtype):
    """36 000 images in one call: the first block's output alone is 5 GB, so clip offsets must be 64-bit.  Batch invariance:
    the last 300 logits equal a 300-image call bit for bit."""
    m = cda.create_model(kind, n_mels=90, compute_dtype=dtype).cuda().eval()
    x = torch.rand(36000, 1, 90, 101, device="cuda")
    y = m(x)
    assert torch.isfinite(y).all() and torch.equal(y[-300:], m(x[-300:]))


@pytest.mark.parametrize("kind", ["standard", "small"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_nan_pixel_gives_nan_logits_as_torch_does(cnn_golden, kind, dtype):
    """torch's ReLU and max-pool propagate NaN, so one NaN pixel makes that clip's logits NaN in the reference (and only that
    clip's); v_max_f32 would have dropped it (nn_common.h: nan_rule_kernel)."""
    sd = cnn_golden[kind][0]
    m = cda.create_model(kind, n_mels=90, compute_dtype=dtype)
    m.load_state_dict(sd)
    m.cuda().eval()
    x = torch.rand(5, 1, 90, 101, generator=torch.Generator().manual_seed(3))
    clean = m(x.cuda()).cpu()
    x[1, 0, 89, 100] = float("nan")
    x[3, 0, 0, 0] = float("nan")
    got = m(x.cuda()).cpu()
    want = ocnn.FORWARD[kind](x, sd)
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.isnan(got[[1, 3]]).all()
    assert torch.equal(got[[0, 2, 4]], clean[[0, 2, 4]])
    preds, probs = m.predict(x.cuda())
    assert torch.isnan(probs[[1, 3]]).all() and preds[1].item() == 0 and preds[3].item() == 0
