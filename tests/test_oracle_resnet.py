"""oracle/resnet.py against goldens produced by the REFERENCE module (oracle/make_golden.py)."""
import torch

from cough_detector_amd import synth
from oracle import resnet


def test_state_dict_keys_match_reference(resnet_golden):
    sd, _ = resnet_golden
    assert set(synth.random_state_dict().keys()) == set(sd.keys())
    for k, v in synth.random_state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    n_params = sum(v.numel() for k, v in sd.items()
                   if "running" not in k and "num_batches" not in k)
    assert n_params == 290370


def test_forward_matches_reference_goldens(resnet_golden):
    sd, vec = resnet_golden
    logits, (a1, a2, a3) = resnet.forward(vec["x"], sd, return_intermediates=True)
    assert a1.shape == (32, 32, 22, 25) and a2.shape == (32, 64, 11, 13) and a3.shape == (32, 128, 6, 7)
    for got, want in ((a1, vec["a1"]), (a2, vec["a2"]), (a3, vec["a3"]), (logits, vec["logits"])):
        assert (got - want).abs().max() < 2e-5
    preds, probs = resnet.predict(vec["x"], sd)
    assert torch.equal(preds, vec["preds"])
    assert (probs - vec["probs"]).abs().max() < 1e-5
    assert set(vec["preds"].tolist()) == {0, 1}
