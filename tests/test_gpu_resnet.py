"""K2-K5 parity: HIP classifier (through the C-ABI) vs reference-generated goldens and the CPU oracle."""
import pytest
import torch

import cough_detector_amd as cda
from cough_detector_amd import synth
from oracle import featurizer as ofeat, resnet as ores
from parity import LOGIT_TOL, SHIPPED, synth_batch

pytestmark = pytest.mark.gpu


def make_model(sd, dtype):
    m = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype)
    m.load_state_dict(sd)
    return m.cuda().eval()


@pytest.mark.parametrize("dtype", ["fp32", "_direct", "bf16x3"])
def test_reference_goldens(resnet_golden, dtype):
    """Reference-generated goldens with a trained-scale head (class-margin std 2.5, |logit| up to 3.9): logits within
    1e-3 abs and argmax exact on EVERY clip -- no margin mask -- for the exact-f32 kernels and for the split-bf16
    ("bf16x3") kernels the benchmark runs."""
    sd, vec = resnet_golden
    m = make_model(sd, dtype)
    logits = m(vec["x"].cuda())
    for which, key in ((1, "a1"), (2, "a2"), (3, "a3")):
        got = m.read_activation(which).cpu()
        assert got.shape == vec[key].shape
        err = (got - vec[key]).abs().max().item()
        print(f"{dtype} {key}: max abs err {err:.2e} (ref max {vec[key].abs().max():.2f})")
        assert err < 2e-5 * max(1.0, vec[key].abs().max().item())
    err = (logits.cpu() - vec["logits"]).abs().max().item()
    print(f"{dtype} logits: max abs err {err:.2e}")
    assert err < (LOGIT_TOL if dtype == "bf16x3" else 1e-4)
    preds, probs = m.predict(vec["x"].cuda())
    assert (probs.cpu() - vec["probs"]).abs().max() < LOGIT_TOL
    assert torch.equal(preds.cpu(), vec["preds"])


def test_plain_bf16_is_approximate(resnet_golden):
    """compute_dtype="bf16" (single bf16 operands and activations) is the fast APPROXIMATE mode: on the trained-scale
    goldens its logit error is ~2 % of the class-margin spread (5.5e-2 measured), far outside LOGIT_TOL, so it is NOT
    what the headline runs.  This pins the size of that error (it must stay a few percent of the margin spread) and
    that classes agree wherever the reference margin exceeds twice the error."""
    sd, vec = resnet_golden
    m = make_model(sd, "bf16")
    logits = m(vec["x"].cuda()).cpu()
    err = (logits - vec["logits"]).abs().max().item()
    spread = (vec["logits"][:, 1] - vec["logits"][:, 0]).std().item()
    print(f"bf16 logits: max abs err {err:.2e} = {100 * err / spread:.1f} % of the margin spread")
    assert LOGIT_TOL < err < 0.05 * spread
    margin = (vec["logits"][:, 1] - vec["logits"][:, 0]).abs()
    keep = margin > 2 * err
    assert torch.equal(logits.argmax(1)[keep], vec["preds"][keep]) and keep.float().mean() > 0.8


def test_single_window_batch1_like_reference_engine(resnet_golden):
    sd, vec = resnet_golden
    m = make_model(sd, "fp32")
    for i in (0, 5, 11):
        out = m(vec["x"][i:i + 1].cuda())                     # (1, 1, 90, 101), inference.py:179-183
        assert out.shape == (1, 2)
        assert (out.cpu() - vec["logits"][i:i + 1]).abs().max() < 5e-5
    cpu_out = m(vec["x"][:2])                                  # CPU tensor in -> CPU tensor out
    assert not cpu_out.is_cuda


@pytest.mark.parametrize("shape", [(103, 101), (90, 51), (64, 101), (40, 33)])
def test_other_input_shapes(shape):
    sd = synth.random_state_dict(seed=11)
    m = make_model(sd, "fp32")
    x = torch.randn(5, 1, *shape, generator=torch.Generator().manual_seed(0))
    ref, inter = ores.forward(x, sd, return_intermediates=True)
    got = m(x.cuda())
    for which in (1, 2, 3):
        assert (m.read_activation(which).cpu() - inter[which - 1]).abs().max() < 5e-5
    assert (got.cpu() - ref).abs().max() < 5e-5


def test_weights_reload_rebuilds_native_handle():
    x = torch.randn(3, 1, 90, 101, generator=torch.Generator().manual_seed(1))
    m = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1).cuda()
    for seed in (1, 2):
        sd = synth.random_state_dict(seed=seed)
        m.load_state_dict(sd)
        assert (m(x.cuda()).cpu() - ores.forward(x, sd)).abs().max() < 5e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_full_size_batch(resnet_golden, dtype):
    """BASELINE configs[2]: B = 4096 end-to-end features + classifier; logits 1e-3 abs, argmax exact."""
    sd, _ = resnet_golden
    B = 4096
    w = synth_batch(5000, B)
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    feats = pre.featurize_batch(w.cuda())
    m = make_model(sd, dtype)
    preds, probs = m.predict(feats.unsqueeze(1))
    logits = m(feats.unsqueeze(1))
    assert torch.isfinite(logits).all()
    # batch invariance: same clips in a small batch give bit-identical logits
    idx = torch.tensor([0, 31, 32, 100, 4095])
    assert torch.equal(m(feats[idx].unsqueeze(1)), logits[idx])
    # softmax / argmax consistency over the whole batch
    assert torch.equal(preds, logits.argmax(dim=1))
    assert (probs - torch.softmax(logits, dim=1)).abs().max() < 1e-6
    # oracle (CPU featuriser + CPU classifier) on a sample
    sample = torch.arange(0, B, 16)
    ref_logits = ores.forward(ofeat.extract_features_batch(w[sample]).unsqueeze(1), sd)
    err = (logits[sample].cpu() - ref_logits).abs()
    print(f"{dtype} B=4096 sample of {len(sample)}: logits max abs err {err.max():.2e}")
    assert err.max() < LOGIT_TOL
    # argmax: exact wherever the reference margin exceeds the tolerance itself (both classes of a tie-within-tolerance
    # clip are "within 1e-3"); with a margin spread of ~2.5 that is every clip but a handful
    margin = (ref_logits[:, 1] - ref_logits[:, 0]).abs()
    keep = margin > 2 * LOGIT_TOL
    assert torch.equal(logits[sample].cpu().argmax(1)[keep], ref_logits.argmax(1)[keep])
    assert keep.float().mean() > 0.98


def test_empty_batch(resnet_golden):
    sd, _ = resnet_golden
    m = make_model(sd, "fp32")
    assert m(torch.zeros(0, 1, 90, 101, device="cuda")).shape == (0, 2)
