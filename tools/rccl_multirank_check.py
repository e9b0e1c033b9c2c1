"""Hardware check of the W > 1 exchange (the ragged last bucket, send-buffer reuse two buckets later and
the flush across the end of the stream have only ever run with one rank or on gloo).  Needs >= 2 GPUs on one node:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/rccl_multirank_check.py

(start it with torchrun: the launcher never touches the GPU).  On a ONE-GPU box the same job can be rehearsed with W ranks
sharing the card -- RCCL refuses two ranks on one device, so the exchange then runs on gloo (CUDA tensors staged through
the host): `COUGH_CHECK_BACKEND=gloo COUGH_CHECK_SHARE_GPU=1 python -m torch.distributed.run ... --nproc-per-node 2 ...`
exercises the W = 2 shard -> generate -> score -> bucketed gather -> un-interleave path with the real kernels; only the
transport differs.  Every rank scores its round-robin shard of a ragged
synthetic stream through CoughPipeline + BucketedLogitsGather on RCCL (`distributed.score_stream`), then scores the
WHOLE stream alone; the gathered logits must be bit-identical to the single-rank ones on every rank.  Not part of
`pytest -m gpu`: the one-GPU boxes cannot run it, and a test process that has initialised the GPU may not start ranks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import cough_detector_amd as cda
from cough_detector_amd import distributed as cdist, synth


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("COUGH_CHECK_BACKEND", "nccl")
    if os.environ.get("COUGH_CHECK_SHARE_GPU") == "1":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    pre = cda.AudioPreprocessor(device="cuda", use_pcen=False, use_pre_emphasis=False, use_delta_delta=False,
                                use_spectral_contrast=False)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16x3")
    model.load_state_dict(synth.random_state_dict(seed=3))
    model.to(dev).eval()
    pipe = cda.CoughPipeline(pre, model)
    ok = True
    for total, batch, every in ((3 * 1024 * world + 517, 1024, 2), (1000, 512, 1), (world - 1 or 1, 256, 4)):
        full = cdist.score_stream(pipe, total, batch=batch, every=every)
        alone = torch.cat([pipe(b, normalize=True) for b in cdist.stream_shard(total, batch, 0, 1, dev)[0] if b.shape[0]])
        same = bool(torch.equal(full, alone))
        ok &= same
        print(f"rank {rank}/{world}: stream of {total} clips, batch {batch}, every {every}: gathered == single-rank: {same}",
              flush=True)
    t = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    if rank == 0:
        print(f"{backend} multi-rank exchange ({world} ranks):", "OK" if t.item() == 1.0 else "MISMATCH")
    sys.exit(0 if t.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
