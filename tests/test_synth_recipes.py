"""The two synthetic clip sources: ``synth.make_clip`` (numpy PCG64 stream; the committed fixtures) and the counter-based
recipe ``synth.make_clip_counter`` whose device twin ``cough_synth_clips`` feeds the 1M-clip benchmark stream
(BASELINE.json configs[3]).  Both follow /root/reference/setup_coughvid.py:381-441, mixture by seed % 6."""
import numpy as np
import pytest
import torch

from cough_detector_amd import synth


def test_counter_recipe_is_deterministic_and_matches_the_numpy_recipe_statistically():
    assert np.array_equal(synth.make_clip_counter(1234), synth.make_clip_counter(1234))
    assert not np.array_equal(synth.make_clip_counter(6), synth.make_clip_counter(12))
    for kind in range(6):
        a = np.stack([synth.make_clip_counter(kind + 6 * k) for k in range(24)])
        b = np.stack([synth.make_clip(kind + 6 * k, peak_normalize=False) for k in range(24)])
        assert a.dtype == np.float32 and a.shape == (24, 16000) and np.isfinite(a).all()
        rms_a, rms_b = np.sqrt((a ** 2).mean()), np.sqrt((b ** 2).mean())
        pk_a, pk_b = np.abs(a).max(axis=1).mean(), np.abs(b).max(axis=1).mean()
        assert 0.7 < rms_a / rms_b < 1.4, (kind, rms_a, rms_b)
        assert 0.7 < pk_a / pk_b < 1.4, (kind, pk_a, pk_b)
    cough = synth.make_clip_counter(0)
    assert 0.75 < np.abs(cough).max() < 0.9                      # burst scaled to 0.8 + floor
    hum = np.abs(np.fft.rfft(synth.make_clip_counter(3)))
    assert hum.argmax() in (50, 60, 100, 120)                    # 1 Hz bins


@pytest.mark.gpu
def test_device_generator_matches_host_mirror():
    """cough_synth_clips (csrc/synth.hip) vs its host mirror, sample by sample: the parameter arithmetic is bit-exact by
    construction; samples differ only by the last bits of sin / cos / log / exp."""
    seeds = list(range(0, 60)) + [999_983, 4_000_000_007]
    got = torch.cat([synth.device_clips(s, 1) for s in seeds]).cpu().numpy()
    want = np.stack([synth.make_clip_counter(s) for s in seeds])
    err = np.abs(got - want).max(axis=1)
    worst = int(err.argmax())
    wi = int(np.abs(got[worst] - want[worst]).argmax())
    print("device generator vs host mirror: max abs diff %.2e (seed %d, kind %d, sample %d: %r vs %r); per kind %s"
          % (err.max(), seeds[worst], seeds[worst] % 6, wi, got[worst, wi], want[worst, wi],
             [float("%.1e" % max(e for e, s in zip(err, seeds) if s % 6 == k)) for k in range(6)]))
    assert err.max() < 2e-5
    # strided seeds = a round-robin shard; rows are independent of the launch they were produced in
    shard = synth.device_clips(3, 16, seed_stride=8).cpu().numpy()
    assert np.array_equal(shard[2], synth.device_clips(19, 1).cpu().numpy()[0])
    assert synth.device_clips(0, 0).shape == (0, 16000)
