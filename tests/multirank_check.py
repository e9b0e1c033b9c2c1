"""Hardware check of the W > 1 exchange, run as torchrun RANKS (never imported by pytest; `tests/test_gpu_multirank.py`
starts it as a fresh child process when the node has >= 2 GPUs).  By hand, on a node with N GPUs:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tests/multirank_check.py

Every rank scores its round-robin shard of three ragged synthetic streams through CoughPipeline + BucketedLogitsGather
on RCCL (`distributed.score_stream`: ragged last bucket, send-buffer reuse two buckets later, the flush across the end
of the stream), then scores the WHOLE stream alone; on every rank the gathered logits must be
  (1) bit-identical to the single-rank ones, and
  (2) within the logit tolerance of the CPU ORACLE at 64 sampled GLOBAL indices (rank 0; the oracle is the checker).
On a ONE-GPU box the same job can be rehearsed with W ranks sharing the card -- RCCL refuses two ranks on one device, so
the exchange then runs on gloo: `COUGH_CHECK_BACKEND=gloo COUGH_CHECK_SHARE_GPU=1 python -m torch.distributed.run ...
--nproc-per-node 2 tests/multirank_check.py`; only the transport differs."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np
import torch
import torch.distributed as dist

import cough_detector_amd as cda
from cough_detector_amd import distributed as cdist, synth
from cough_detector_amd.hostcpu import bind_to_gpu_numa, bound_torch_threads


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("COUGH_CHECK_BACKEND", "nccl")
    if os.environ.get("COUGH_CHECK_SHARE_GPU") == "1":
        local = local % max(torch.cuda.device_count(), 1)
    numa = bind_to_gpu_numa(local)
    bound_torch_threads()
    if torch.cuda.device_count() <= local:
        print(f"rank {rank}: only {torch.cuda.device_count()} device(s) visible, need device {local}", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    from parity import LOGIT_TOL, realistic_state_dict           # tests/parity.py: trained-scale head
    sd = realistic_state_dict(11)
    pre = cda.AudioPreprocessor(device="cuda", use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                                use_spectral_contrast=False)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(sd)
    model.to(dev).eval()
    pipe = cda.CoughPipeline(pre, model)
    ok = True
    for total, batch, every in ((3 * 1024 * world + 517, 1024, 2), (1000, 512, 1), (world - 1 or 1, 256, 4)):
        full = cdist.score_stream(pipe, total, batch=batch, every=every)
        alone = torch.cat([pipe(b, normalize=True) for b in cdist.stream_shard(total, batch, 0, 1, dev)[0] if b.shape[0]])
        same = bool(torch.equal(full, alone))
        msg = f"rank {rank}/{world} (device {local}, numa {numa['numa_node']}): stream of {total} clips, batch {batch}, " \
              f"every {every}: gathered == single-rank: {same}"
        if rank == 0:                                             # the CPU oracle at sampled global indices
            from oracle import featurizer as ofeat, resnet as ores
            rng = np.random.default_rng(total)
            idx = np.unique(np.concatenate([[0, total - 1], rng.integers(0, total, 62)]))
            wav = torch.from_numpy(np.stack([synth.make_clip_counter(int(g)) for g in idx]))
            ref = ores.forward(ofeat.extract_features_batch(wav, normalize_first=True).unsqueeze(1), sd)
            got = full[torch.from_numpy(idx).to(dev)].cpu()
            err = (got - ref).abs().max().item()
            amax = bool(torch.equal(got.argmax(1), ref.argmax(1)))
            same &= err < LOGIT_TOL and amax
            msg += f"; vs CPU oracle at {len(idx)} global indices: max|dlogit| {err:.2e} (< {LOGIT_TOL}), argmax equal: {amax}"
        ok &= same
        print(msg, flush=True)
    t = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    if rank == 0:
        print(f"{backend} multi-rank exchange ({world} ranks):", "OK" if t.item() == 1.0 else "MISMATCH")
    sys.exit(0 if t.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
