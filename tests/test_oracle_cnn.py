"""oracle/cnn.py against goldens produced by the REFERENCE modules CoughDetector / CoughDetectorSmall
(oracle/make_golden_cnn.py)."""
import pytest
import torch

from oracle import cnn


@pytest.mark.parametrize("kind,n_params,conv_shape", [("standard", 421954, (8, 256, 5, 6)), ("small", 21122, (8, 128, 11, 12))])
def test_forward_matches_reference_goldens(cnn_golden, kind, n_params, conv_shape):
    sd, vec = cnn_golden[kind]
    x = cnn_golden["x"]
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == n_params
    conv_out = cnn.FEATURES[kind](x, sd)
    assert tuple(conv_out.shape) == conv_shape
    assert (conv_out - vec["conv_out"]).abs().max() < 2e-5
    logits = cnn.FORWARD[kind](x, sd)
    assert (logits - vec["logits"]).abs().max() < 1e-4          # trained-scale head: |logit| up to 8
    preds, probs = cnn.predict(kind, x, sd)
    assert torch.equal(preds, vec["preds"]) and set(preds.tolist()) == {0, 1}
    assert (probs - vec["probs"]).abs().max() < 1e-5
