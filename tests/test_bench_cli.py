"""bench.py's command-line plumbing (no GPU): the multi-rank parent path and its error message."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_launch_command_is_torchrun_with_forwarded_arguments():
    argv = ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    cmd = bench.build_launch_cmd(argv, 4, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv


def test_defaults_match_the_driver_contract():
    a = bench.parse_args([])
    assert a.gpus == 1 and a.dtype == "bf16x3" and a.rotate >= 3 and a.prewarm_s >= 0.5 and a.total_clips == 0
    assert a.rotate * a.batch * 64000 > 2 * 256 * 2**20        # rotation set > 2 x the Infinity Cache


def test_more_gpus_than_devices_fails_with_a_clear_message():
    """`python bench.py --gpus 2` without a torchrun environment is the PARENT: it must say how many devices it sees
    (this container: 0; a one-GPU box: 1) and exit non-zero without touching the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and p.stdout == ""
    assert "--gpus 64: only" in p.stderr and "device(s) visible" in p.stderr
