"""bench.py's command-line plumbing (no GPU): the multi-rank parent path and its error message."""
import os
import subprocess
import sys

import pytest

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


def test_explicit_visible_devices_shortage_fails_with_a_clear_message():
    """`python bench.py --gpus 2` without a torchrun environment is the PARENT.  A *_VISIBLE_DEVICES variable that names
    fewer devices than --gpus is explicit evidence: it must say so and exit 2 without touching the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2 and p.stdout == ""
    assert "--gpus 2: only 1 device(s) visible" in p.stderr and "HIP_VISIBLE_DEVICES" in p.stderr


def test_sysfs_shortage_is_only_a_warning_and_the_ranks_are_launched(monkeypatch, capsys):
    """the sysfs device count is a heuristic; a false negative in an unfamiliar container must not
    void an 8-GPU run.  Without an explicit *_VISIBLE_DEVICES limit the parent warns and starts torchrun anyway."""
    from cough_detector_amd import hostcpu
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(hostcpu, "visible_gpu_count", lambda *a, **k: 0)
    started = {}

    class Done:
        returncode = 0
        stdout = 'NCCL version banner\n{"metric": "x", "value": 1}\n'

    def fake_run(cmd, **kw):
        started["cmd"], started["env"] = cmd, kw["env"]
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    args = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    rc = bench.launch_ranks(args, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    out = capsys.readouterr()
    assert rc == 0 and "--nproc-per-node=8" in started["cmd"] and started["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "WARNING: sysfs shows only 0 device(s)" in out.err and "NCCL version banner" in out.err
    assert out.out.strip() == '{"metric": "x", "value": 1}'                  # exactly rank 0's JSON line on stdout


def test_unknown_env_dtype_is_rejected_and_the_old_name_is_mapped(monkeypatch, capsys):
    """argparse does not validate an environment default against `choices`."""
    monkeypatch.setenv("COUGH_BENCH_DTYPE", "bf16")
    assert bench.parse_args([]).dtype == "bf16_approx" and "APPROXIMATE" in capsys.readouterr().err
    monkeypatch.setenv("COUGH_BENCH_DTYPE", "fp8")
    with pytest.raises(SystemExit):
        bench.parse_args([])
