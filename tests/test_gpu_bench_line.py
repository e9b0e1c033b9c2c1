"""The driver's contract with bench.py, checked in-process on the GPU (no child process): ONE JSON line with the fields
the prompt's measurement section names -- metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload, a `roofline` object for the dominant kernel and a
`cpu_baseline` object -- at a small batch so that it runs in seconds."""
import json
import sys

import pytest

pytestmark = pytest.mark.gpu


def _run(monkeypatch, capsys, *argv):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("COUGH_BENCH_FORCE_DIST", raising=False)
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_default_line_has_every_contract_field(monkeypatch, capsys):
    d = _run(monkeypatch, capsys, "--gpus", "1", "--steps", "6", "--warmup", "2", "--batch", "512", "--cpu-seconds", "1.0",
             "--prewarm-s", "0.05", "--no-live-pmc")
    assert d["metric"].startswith("1s@16kHz clips/sec") and d["unit"] == "clips/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and d["dtype"].startswith("bf16x3") and "workload" in d["config"]
    assert "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert r["algorithmic_bytes_per_launch"] == 512 * 100360                   # 64 000 read + 36 360 written
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["ms_per_launch"] < d["ms_per_step"]                               # the dominant kernel fits inside the step
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c
    assert c["kind"] == "port" and c["unit"] == "clips/s" and c["value"] > 0 and c["cores"] >= 1
    cl = d["roofline_classifier"]
    assert cl["bound"] == "mfma" and cl["peak"] == 2500.0 and 0 < cl["frac"] < 1 and cl["mfma_per_product"] == 3
    s = d["roofline_stft"]
    assert s["bound"] == "hbm" and s["algorithmic_bytes_per_launch"] == 512 * 167828 and 0 < s["frac"] < 1


def test_short_total_clips_stream_does_not_crash(monkeypatch, capsys):
    """a --total-clips stream shorter than one batch has no full step; the line must still come out."""
    d = _run(monkeypatch, capsys, "--total-clips", "300", "--batch", "512", "--cpu-seconds", "0", "--prewarm-s", "0.05")
    assert d["scaling"] == "strong" and d["steps"] == 1 and d["value"] > 0
    assert d["roofline"]["clips_per_launch"] == 300 and "roofline_stft" not in d


def test_roofline_traffic_is_collected_live_with_rocprofv3(monkeypatch, capsys):
    """roofline.traffic of the default run comes from two `rocprofv3 --pmc` child runs of bench.py itself (FETCH_SIZE, WRITE_SIZE,
    separate passes), not from a committed file: for the fused featurise + stem kernel at batch 512 it must sit at 1.34x the
    SURVEY 8d bytes (the f32 a1 hand-off) within a few per cent."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    d = _run(monkeypatch, capsys, "--steps", "4", "--warmup", "1", "--batch", "512", "--cpu-seconds", "0", "--prewarm-s", "0.05")
    r = d["roofline"]
    if not r["traffic_source"].startswith("live: rocprofv3 --pmc"):
        # counters are a property of the box (another profiler session, restricted perf counters): bench.py then falls back to
        # the committed record and says so -- an environment condition, not a defect of the path under test
        pytest.skip(f"rocprofv3 --pmc delivered no counters on this box; bench.py fell back to: {r['traffic_source']}")
    assert 1.25 < r["traffic"] / (512 * 100360) < 1.45 and abs(r["traffic_over_algorithmic"] - r["traffic"] / (512 * 100360)) < 2e-3
    rs = d["roofline_stft"]
    assert rs["traffic_source"].startswith("live: rocprofv3 --pmc") and 0.97 < rs["traffic"] / (512 * 167828) < 1.08
    busy = d["roofline_classifier"]["mfma_pipe_busy"]
    assert 0.05 < busy["block0"] < 1.0 and 0.05 < busy["block1"] < 1.0


def test_streaming_bench_prints_one_aggregated_line(monkeypatch, capsys):
    """bench_streaming.py (configs[4]): ONE JSON line for the job -- p50 as `value`, global percentiles, aggregate windows/s,
    one record per rank -- in-process at 8 streams x 4 s (the N > 1 aggregation is covered by gloo tests on CPU)."""
    import bench_streaming
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("COUGH_BENCH_FORCE_DIST", raising=False)
    assert bench_streaming.main(["--streams", "8", "--seconds", "4"]) == 0
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["unit"] == "ms" and d["higher_is_better"] is False and d["n_gpus"] == d["rccl_world"] == 1
    assert d["value"] == d["latency_ms_p50"] and 0 < d["latency_ms_p50"] <= d["latency_ms_p99"] <= d["latency_ms_max"] < 50
    assert d["config"]["streams"] == 8 and len(d["ranks"]) == 1 and d["ranks"][0]["streams"] == 8
    assert d["windows"] == 8 * 13 == d["ranks"][0]["windows"]                     # (4 s - 1 s) / 0.25 s + 1 windows per stream
    assert d["sustained_windows_per_s"] > d["real_time_need_windows_per_s"] == 32.0
