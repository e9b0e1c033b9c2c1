"""configs[3] at N = 1 through the REAL multi-rank code, and the CLI loop of run_detection.py.

(a) A ~20 k-clip stream with a ragged tail goes through ``distributed.score_stream``: round-robin shard -> on-device
    clip generator (``cough_synth_clips``, the recipe of /root/reference/setup_coughvid.py:381-441) ->
    ``CoughPipeline`` (bf16x3) -> ``BucketedLogitsGather`` on a ONE-RANK RCCL process group (backend "nccl", built in
    this process: no child, no exec) -> un-interleave.  The gathered logits at sampled GLOBAL indices must equal the
    CPU oracle on ``synth.make_clip_counter(index)``.  A second case shards the same stream as rank r of W = 3 without
    a group: local row j must be global clip r + 3 j.
(b) ``inference.main([...])`` -- what ``run_detection.py`` calls (/root/reference/run_detection.py:14-17 ->
    src/inference.py:454-503) -- on a saved reference-schema checkpoint and a ``.npy`` stream, against
    ``oracle/engine.py``.
"""
import os
import socket

import numpy as np
import pytest
import torch

import cough_detector_amd as cda
from cough_detector_amd import distributed as cdist, inference, synth
from oracle import engine as oengine, featurizer as ofeat, resnet as ores
from parity import LOGIT_TOL, realistic_state_dict

pytestmark = pytest.mark.gpu

SHIPPED = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)
CONFIG = dict(model_type="residual", sample_rate=16000, n_mels=64, n_fft=512, hop_length=160, win_length=400,
              f_min=100.0, f_max=4000.0, segment_duration=1.0, n_mfcc=13, use_mfcc=True, use_pcen=False,
              use_pre_emphasis=False, pre_emphasis_coef=0.97, use_delta_delta=False, use_spectral_contrast=False,
              n_contrast_bands=6)


def _pipeline(sd, dtype="bf16x3"):
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=dtype)
    model.load_state_dict(sd)
    model.to("cuda").eval()
    return cda.CoughPipeline(pre, model)


def _oracle_logits(sd, indices):
    wav = torch.from_numpy(np.stack([synth.make_clip_counter(int(g)) for g in indices]))
    feats = ofeat.extract_features_batch(wav, normalize_first=True)
    return ores.forward(feats.unsqueeze(1), sd)


@pytest.fixture()
def one_rank_rccl():
    """A world-size-1 process group on the RCCL backend inside this process (the same code path the ranks of
    ``bench.py --gpus N`` take: ``all_gather_into_tensor`` on RCCL's stream, async work handles)."""
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_configs3_stream_through_rccl_gather_matches_oracle_at_global_indices(one_rank_rccl):
    sd = realistic_state_dict(11)
    pipe = _pipeline(sd)
    total, batch, every = 4 * 4096 + 3653, 4096, 2        # 5 steps (ragged last), 3 exchanges (last one partial)
    full = cdist.score_stream(pipe, total, batch=batch, every=every)
    assert one_rank_rccl.get_backend() == "nccl" and full.shape == (total, 2)
    rng = np.random.default_rng(0)
    idx = np.unique(np.concatenate([[0, 1, batch - 1, batch, 2 * batch - 1, 2 * batch, 4 * batch, total - 1],
                                    rng.integers(0, total, 56)]))
    ref = _oracle_logits(sd, idx)
    got = full[torch.from_numpy(idx).cuda()].cpu()
    assert (got - ref).abs().max().item() < LOGIT_TOL
    assert torch.equal(got.argmax(1), ref.argmax(1))
    # the same stream without the exchange: the bucketed gather must be a pure re-ordering (bit-identical)
    plain = torch.cat([pipe(b, normalize=True) for b in cdist.stream_shard(total, batch, 0, 1, "cuda")[0]])
    assert torch.equal(plain, full)


@pytest.mark.parametrize("rank,world", [(1, 3), (7, 8)])
def test_configs3_shard_of_rank_r_holds_global_clips_r_plus_jW(rank, world):
    sd = realistic_state_dict(11)
    pipe = _pipeline(sd)
    total, batch = 9001, 1024
    local = cdist.score_stream(pipe, total, batch=batch, rank=rank, world=world)     # no group: this rank's rows
    n_local = cdist.local_count(total, rank, world)
    assert local.shape == (n_local, 2)
    js = np.unique(np.concatenate([[0, 1, n_local - 1], np.random.default_rng(rank).integers(0, n_local, 13)]))
    ref = _oracle_logits(sd, rank + js * world)
    got = local[torch.from_numpy(js).cuda()].cpu()
    assert (got - ref).abs().max().item() < LOGIT_TOL and torch.equal(got.argmax(1), ref.argmax(1))


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_cli_main_detections_match_engine_oracle(tmp_path, capsys, dtype):
    # trained-scale head (class-margin std 2.5): a default-init head (margin spread ~0.01) would hide any reduced-precision
    # error of the conv stack behind a near-degenerate Linear layer and the 1e-3 bound below would say nothing about bf16x3
    sd = realistic_state_dict(5)
    ckpt = str(tmp_path / "best_model.pt")
    torch.save({"epoch": 3, "model_state_dict": sd, "optimizer_state_dict": {}, "metrics": {"f1": 0.5},
                "config": CONFIG}, ckpt)                                 # schema of src/train.py:192-198
    stream = synth.make_stream(9, 6.0)
    npy = str(tmp_path / "stream.npy")
    np.save(npy, stream)
    out = inference.main(["--model", ckpt, "--threshold", "0.5", "--smoothing", "3", "--debounce", "0.5",
                          "--input", npy, "--compute-dtype", dtype])
    printed = capsys.readouterr().out
    now = {"t": 0.0}
    ref = oengine.EngineOracle(sd, 0.5, 3, 0.5, clock=lambda: now["t"])
    ref_hits = []
    for i in range(0, len(stream) - 1600 + 1, 1600):                    # 0.1 s chunks, inference.py:259,275
        now["t"] = (i + 1600) / 16000.0
        hit = ref.process_audio_chunk(stream[i:i + 1600])
        if hit is not None:
            ref_hits.append(hit)
    assert len(out["window_probs"]) == len(ref.window_probs) == 21
    probs = np.array(ref.window_probs)
    assert probs.max() - probs.min() > 0.3                               # the head separates the windows of this stream
    assert np.abs(np.array(out["window_probs"]) - probs).max() < 1e-3
    assert len(ref_hits) >= 2 and len(out["detections"]) == len(ref_hits)
    for (t, conf), (rt, rconf) in zip(out["detections"], ref_hits):
        assert t == pytest.approx(rt) and abs(conf - rconf) < 1e-3
    assert printed.count("COUGH DETECTED") == len(ref_hits) and "Model loaded: residual" in printed


def test_cli_main_flags_of_the_reference_are_accepted(tmp_path, capsys):
    """Every flag of src/inference.py:455-475 parses; --list-devices returns without loading a model."""
    assert inference.main(["--model", "unused.pt", "--list-devices"]) is None
    assert "No audio capture back-end" in capsys.readouterr().out
    sd = synth.random_state_dict(seed=5)
    ckpt = str(tmp_path / "m.pt")
    torch.save({"model_state_dict": sd, "config": CONFIG}, ckpt)
    out = inference.main(["--model", ckpt, "--threshold", "0.99", "--smoothing", "5", "--debounce", "1.0",
                          "--device", "cuda", "--audio-device", "2", "--backend", "pyaudio", "--quiet",
                          "--seconds", "2.0"])
    assert capsys.readouterr().out == "" and len(out["window_probs"]) == 5
    with pytest.raises(ValueError, match="MI355X only"):
        inference.main(["--model", ckpt, "--device", "cpu", "--quiet"])
