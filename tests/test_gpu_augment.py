"""SpecAugment on the HIP path against the CPU restatement (oracle/augmentation.py), same seeded host RNG."""
import random

import pytest
import torch

from cough_detector_amd.augmentation import SpecAugment
from oracle import augmentation as oaug

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 90, 101), (7, 1, 90, 101), (3, 2, 64, 33)])
def test_masks_equal_the_oracle_for_the_same_seeds(shape):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(shape, generator=g) + 0.1                       # strictly positive: zeros are masks
    aug = SpecAugment(freq_mask_param=10, time_mask_param=20, n_freq_masks=2, n_time_masks=2, p=0.7)
    fired = 0
    for seed in range(12):
        random.seed(seed); torch.manual_seed(seed)
        want = oaug.spec_augment(x, 10, 20, 2, 2, 0.7)
        random.seed(seed); torch.manual_seed(seed)
        got = aug(x.cuda())
        assert got.shape == x.shape
        assert torch.equal(got.cpu(), want)                         # bit-exact: masked_fill(0) or pass-through
        fired += int(not torch.equal(want, x))
    assert 0 < fired < 12                                           # both branches of the coin were exercised


def test_properties_on_a_full_batch():
    x = torch.rand((4096, 1, 90, 101), device="cuda") + 0.1
    aug = SpecAugment(p=1.0)
    random.seed(1); torch.manual_seed(1)
    y = aug(x)
    zero = y == 0
    assert torch.equal(y[~zero], x[~zero])                          # untouched elsewhere
    assert torch.equal(zero[0], zero[-1])                           # same masks for every item (iid_masks=False)
    rows = zero[0, 0].all(dim=1).sum().item()
    cols = zero[0, 0].all(dim=0).sum().item()
    assert rows < 2 * 10 and cols < 2 * 20                          # each mask is narrower than its parameter
    assert zero[0, 0].sum().item() == rows * 101 + cols * 90 - rows * cols
    assert aug(x[:0]).shape == (0, 1, 90, 101)
    assert SpecAugment(p=0.0)(x) is x                               # coin says no: the input itself, as the reference
