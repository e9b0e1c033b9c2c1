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


def test_live_pmc_never_nests_profilers_and_scrubs_the_child_environment(monkeypatch):
    """under `rocprofv3 ... -- python bench.py` the process inherits the profiler's LD_PRELOAD /
    ROCP_TOOL_LIBRARIES / ROCPROF_* variables; starting `rocprofv3` again from there would initialise the GPU in the launcher and
    then exec.  live_pmc must (a) start nothing when it already runs under a profiler and say why, (b) otherwise hand its
    children an environment without any profiler or torchrun variable."""
    started = []

    class FakeProc:
        def __init__(self, cmd, **kw):
            started.append((cmd, kw))

        def wait(self, timeout=None):
            return 1                                  # "rocprofv3 failed": the fallback path, with the reason recorded

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(bench.shutil if hasattr(bench, "shutil") else __import__("shutil"), "which", lambda name: "/usr/bin/" + name)
    args = bench.parse_args(["--steps", "3"])
    for k in list(os.environ):
        if k.startswith(("ROCP_", "ROCPROF_", "ROCPROFILER_")) or k in ("LD_PRELOAD", "HSA_TOOLS_LIB"):
            monkeypatch.delenv(k)
    # (a) under a profiler: nothing is started
    for var, val in (("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so"), ("ROCP_TOOL_LIBRARIES", "librocprofiler-sdk-tool.so"),
                     ("ROCPROF_COUNTER_COLLECTION", "1")):
        monkeypatch.setenv(var, val)
        assert bench.under_profiler()
        res, why = bench.live_pmc(args)
        assert res is None and "under a profiler" in why and started == []
        monkeypatch.delenv(var)
    assert not bench.under_profiler()
    # (b) clean run: one child per pass would start; its environment is scrubbed
    monkeypatch.setenv("LD_PRELOAD", "/some/unrelated.so")           # not a profiler, still never handed to a profiled child
    monkeypatch.setenv("WORLD_SIZE", "1")
    res, why = bench.live_pmc(args)
    assert res is None and "exit code 1" in why and len(started) == 1   # first pass failed -> fallback, reason kept
    cmd, kw = started[0]
    assert cmd[0] == "rocprofv3" and "--pmc" in cmd and "--no-live-pmc" in cmd and cmd[cmd.index("--") + 1] == sys.executable
    assert not any(t in cmd for t in ("--kernel-trace", "--sys-trace", "--hip-trace", "-s", "-r"))   # counters only
    env = kw["env"]
    assert "LD_PRELOAD" not in env and "WORLD_SIZE" not in env
    assert not any(k.startswith(("ROCP_", "ROCPROF_")) for k in env) and env["TMPDIR"] == "/tmp"


def test_streaming_bench_parent_command_and_single_aggregated_line():
    """configs[4] under torchrun must yield ONE JSON line for the job -- global latency percentiles over every
    rank's samples, aggregate windows/s over the slowest rank's wall time, the per-rank records -- not one line per rank."""
    import json
    import bench_streaming as bs
    argv = ["--gpus", "8", "--streams", "64", "--seconds", "5"]
    cmd = bs.build_launch_cmd(argv, 8, 29611)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-len(argv):] == argv
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench_streaming.py")
    args = bs.parse_args(argv)
    lat = [[0.10, 0.20, 0.30], [0.40, 5.0], []]                         # rank 2 served no window
    recs = [{"rank": r, "device": r, "streams": s, "ticks": 50, "ticks_with_windows": len(l), "windows": w, "wall_s": t}
            for r, (l, s, w, t) in enumerate(zip(lat, (22, 21, 21), (66, 42, 0), (1.0, 2.0, 0.5)))]
    line = bs.aggregate(lat, recs, args, "gloo", 3)
    json.dumps(line)                                                     # serialisable as it stands
    assert line["n_gpus"] == line["rccl_world"] == 3 and line["config"]["streams"] == 64
    assert line["latency_ms_max"] == 5.0 and line["latency_ms_p50"] == 0.3 and line["value"] == 0.3 and line["unit"] == "ms"
    assert line["windows"] == 108 and line["sustained_windows_per_s"] == 54.0      # 108 windows / the slowest rank's 2.0 s
    assert line["higher_is_better"] is False and len(line["ranks"]) == 3
    assert line["ranks"][2]["latency_ms_p50"] is None and line["ranks"][0]["latency_ms_max"] == 0.3
    # explicit evidence of too few devices: the parent refuses without touching the GPU
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_streaming.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2 and p.stdout == "" and "only 1 device(s) visible" in p.stderr
