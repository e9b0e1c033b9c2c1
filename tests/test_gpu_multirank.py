"""The W > 1 RCCL exchange on real hardware -- runs by itself whenever `pytest -m gpu` lands on a node with >= 2 GPUs.

Skipped on a one-GPU box.  Otherwise a FRESH CHILD process (`python -m torch.distributed.run --nproc-per-node <visible
GPUs, at most 4>` on tests/multirank_check.py; a child, never an exec of this pytest process, which has initialised the
GPU) scores three ragged streams with one rank per GPU -- ragged buckets, send-buffer reuse and the end-of-stream flush
on RCCL over xGMI -- and must exit 0: gathered logits bit-identical to the single-rank ones on every rank and within
the logit tolerance of the CPU oracle at 64 sampled global indices.  SURVEY.md 8e, BASELINE.json configs[3]."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def launch_cmd(n_ranks: int, port: int):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "multirank_check.py")]


def test_all_visible_gpus_rccl_exchange_in_a_fresh_child():
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU visible: the W > 1 RCCL exchange needs >= 2 (it runs by itself on a multi-GPU node)")
    # at most 4 ranks by default: with this pytest process that is 5 processes holding a GPU, inside the 6 a shared box allows a
    # job; COUGH_TEST_MAX_RANKS=8 widens it on a node of one's own
    n = min(n, int(os.environ.get("COUGH_TEST_MAX_RANKS", "4")))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    p = subprocess.run(launch_cmd(n, port), env=env, capture_output=True, text=True, timeout=900)
    tail = (p.stdout + "\n" + p.stderr)[-4000:]
    assert p.returncode == 0, tail
    assert f"nccl multi-rank exchange ({n} ranks): OK" in p.stdout, tail
