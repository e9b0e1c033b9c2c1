#!/usr/bin/env python3
"""Headline benchmark: 1 s @ 16 kHz clips/s through featurise (K1) + CoughDetectorResidual (K2-K5).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment this process only LAUNCHES the ranks (``python -m torch.distributed.run
--nproc-per-node N bench.py ...`` as a child, before anything touches the GPU), relays rank 0's single JSON line
and exits with the children's status; started under torchrun (WORLD_SIZE set) it is one of the ranks.

A step is one pass of the hot path over one batch of B = 4096 synthetic clips that are already resident in HBM
(BASELINE.json configs[2]).  The clips are generated ON THE DEVICE from their global index (``cough_synth_clips``) and
the steps rotate over ``--rotate`` distinct batches (default 3 x 262 MB > 2 x the 256 MiB Infinity Cache), so every
step's waveform read comes from HBM.  With N ranks the clip stream is sharded round-robin (clip i -> rank i mod N,
weak scaling: every rank runs B clips per step) and the only exchange is an RCCL all-gather of the ranks' logits, one
per bucket of ``--gather-every`` steps (default 8: 256 KB per rank), overlapped with the next bucket's compute and
finished inside the timed region.  ``--total-clips T`` runs configs[3] as worded: a T-clip stream (default use: 1 000 000), every clip distinct,
sharded round-robin and resident in HBM (64 KB per clip), one pass.  Rank 0 prints ONE JSON line.

Before ``--warmup`` an untimed pre-warm (``--prewarm-s``, default 0.6 s of the same pipeline, reported as
``prewarm_s``) brings the GPU to its sustained clocks, so a short timed region (the driver's 20 steps = 14 ms) reads
the same as a long one.

Extra objects on that line: ``roofline`` (featurise kernel vs HBM: algorithmic bytes per clip over the kernel's
HIP-event time inside the timed region; ``traffic`` = HBM-side bytes per launch COLLECTED IN THIS RUN by child runs of
this script under ``rocprofv3 --pmc`` (FETCH_SIZE; WRITE_SIZE; MFMA busy -- three passes, ~6 s; ``--no-live-pmc`` or a missing
rocprofv3 falls back to the committed record and says so in ``traffic_source``)), ``roofline_classifier`` (algorithmic 35.67 MFLOP/clip of the residual blocks
vs the dense bf16 MFMA peak; the split-bf16 scheme issues 3 MFMAs per algorithmic one, reported as ``mfma_issue_frac``),
``roofline_stft`` (the STFT stage on its own) and ``cpu_baseline`` (the torch-CPU oracle timed on this box's host
cores on a bounded sample; rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_CLIP = 64000 + 36360          # featurise: waveform read + (90,101) f32 written
BYTES_PER_CLIP_STFT = 64000 + 103828    # STFT stage alone: waveform read + (257,101) f32 power written
# featurise + stem fused: waveform read + the stem's (22,25,32) output written -- bf16 (plain bf16) or f32 (bf16x3)
BYTES_PER_CLIP_FUSED = {"bf16_approx": 64000 + 35200, "bf16x3": 64000 + 70400}
FLOP_PER_CLIP = 42865600                # 2 * 21 432 800 MAC of the classifier
FLOP_PER_CLIP_NO_STEM = 35668480        # minus the stem's 32*45*51*49 MAC (it runs inside the featurise kernel)
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16_approx": 2500.0, "bf16x3": 2500.0, "fp32": 157.3}
MFMA_PER_PRODUCT = {"bf16_approx": 1, "bf16x3": 3, "fp32": 1}
DTYPE_LABEL = {
    "bf16x3": "bf16x3: split-bf16 MFMA (feature image, activations and BN-folded weights as hi+lo bf16 pairs, 3 "
              "v_mfma_f32_32x32x16_bf16 per k-step, f32 accumulate, f32 activations in HBM); featuriser f32",
    "bf16_approx": "bf16 (approximate mode): single-bf16 MFMA operands incl. the stem's feature image, bf16 activations, "
            "f32 accumulate; featuriser f32",
    "fp32": "f32: exact-f32 MFMA (v_mfma_f32_32x32x2_f32); featuriser f32",
}
SHIPPED = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)


def measured_pmc(batch: int, variant: str) -> dict:
    """The committed rocprofv3 PMC record of the K1 launch (tools/pmc_k1.sh -> tools/pmc_to_json.py: HBM-side bytes
    with FETCH_SIZE doubled per the gfx950 calibration, and the VALU-issue share from the SQ pass).  PMC cannot be
    collected from inside this process, so the figures are read from profiles/ and only used when the batch matches.
    ``variant``: "k1" (featurise alone), "k1_fused_bf16_approx" or "k1_fused_bf16x3"."""
    names = {"k1": ["r01_k1_pmc.json"], "k1_fused_bf16_approx": ["r01_k1_fused_pmc.json"],
             "k1_fused_bf16x3": ["r05_k1_fused_x3_pmc.json", "r04_k1_fused_x3_pmc.json", "r03_k1_fused_x3_pmc.json", "r02_k1_fused_x3_pmc.json"]}[variant]
    for name in names:                                   # newest record first
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                p = json.load(f)
            if p["clips_per_launch"] == batch:
                p["source"] = os.path.relpath(path, ROOT)
                return p
        except (OSError, KeyError, ValueError):
            pass
    return {}


LIVE_PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
LIVE_PMC_KERNELS = ("featurize_kernel", "stft3_kernel", "resblock_x3_kernel<32", "resblock_x3_kernel<64")


def under_profiler(env=None) -> bool:
    """True when this process already runs under rocprofv3 / a rocprofiler tool library: the profiler's preloaded library has
    initialised the GPU before main(), and everything it exported (LD_PRELOAD, ROCP_TOOL_LIBRARIES, ROCPROF_*) would be
    inherited by a child."""
    env = os.environ if env is None else env
    if any("rocprof" in env.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return True
    return any(k.startswith("ROCPROF_") or k.startswith("ROCPROFILER_") for k in env)


def clean_child_env(env=None) -> dict:
    """The environment of a child run: this process's, minus everything a profiler exports (so that a nested `rocprofv3`
    starts from a clean slate) and minus the torchrun variables."""
    env = dict(os.environ if env is None else env)
    for k in list(env):
        if k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "WORLD_SIZE", "RANK", "LOCAL_RANK") or k.startswith(("ROCP_", "ROCPROF_", "ROCPROFILER_")):
            del env[k]
    env["TMPDIR"] = "/tmp"
    return env


def run_child(cmd, env, timeout_s: float, stderr_path: str) -> str:
    """Run one child to completion.  Returns "" on success, else the reason.  On a time-out the child gets SIGTERM and a grace
    period before SIGKILL: `rocprofv3` exec's into the profiled program, so the signal lands on a process that is mid-kernel
    under counter collection."""
    with open(stderr_path, "wb") as err:
        try:
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=err)
        except OSError as exc:
            return f"could not start {cmd[0]}: {exc}"
        try:
            rc = proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            proc.terminate()
            try:
                proc.wait(timeout=15)
            except subprocess.TimeoutExpired:
                proc.kill()
                proc.wait()
            return f"timed out after {timeout_s:.0f} s"
    if rc != 0:
        try:
            with open(stderr_path, "rb") as f:
                tail = f.read()[-300:].decode("utf-8", "replace").strip().replace("\n", " | ")
        except OSError:
            tail = ""
        return f"exit code {rc}: {tail}"
    return ""


def live_pmc(args, passes=LIVE_PMC_PASSES, kernels=LIVE_PMC_KERNELS):
    """Hardware counters of this command's kernels, COLLECTED NOW: one child run of this script per pass under
    ``rocprofv3 --pmc <counters>`` (separate passes, counters only, no tracing; the program itself after ``--``).  Children,
    never an exec: this process has initialised the GPU.  Returns ``(result, reason)``: ``{kernel: {counter: mean per launch,
    "launches": n}}`` for the kernels that ran and ``""``, or ``None`` and why (no rocprofv3, already under a profiler, a pass
    failed or timed out) -- the caller then falls back to the committed records and puts the reason into ``traffic_source``.
    Never nests profilers: when this process itself runs under rocprofv3 (``rocprofv3 ... -- python bench.py``) nothing is
    started -- the outer profiler's preload would initialise the GPU inside the inner ``rocprofv3`` launcher, whose exec into the
    program is then the replacement of a GPU-initialised process this pool forbids.  Derived figures (MI355X_MICROARCH.md):
    HBM-side bytes = FETCH_SIZE[KiB] * 1024 * 2 (gfx950 counts half the bytes of a coalesced stream) + WRITE_SIZE[KiB] * 1024;
    MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)."""
    import csv
    import glob
    import shutil
    import tempfile
    if under_profiler():
        return None, "not collected: this run is itself under a profiler (no nested rocprofv3)"
    if shutil.which("rocprofv3") is None:
        return None, "not collected: rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--prewarm-s", "0",
             "--batch", str(args.batch), "--dtype", args.dtype, "--no-live-pmc", "--stft-launches", "4"] + \
            (["--featurize-only"] if args.featurize_only else [])
    res = {}
    tmp = tempfile.mkdtemp(prefix="cough_pmc_", dir="/tmp")
    env = clean_child_env()
    try:
        for n, counters in enumerate(passes):
            out = os.path.join(tmp, f"pass{n}")
            why = run_child(["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", out, "--", *child], env, 90.0,
                            os.path.join(tmp, f"pass{n}.err"))
            if why:
                return None, f"not collected: rocprofv3 --pmc {' '.join(counters)}: {why}"
            sums = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get("Counter_Name") not in counters:
                            continue
                        for k in kernels:
                            if k in r.get("Kernel_Name", ""):
                                acc = sums.setdefault((k, r["Counter_Name"]), [0.0, 0])
                                acc[0] += float(r["Counter_Value"])
                                acc[1] += 1
            for (k, c), (tot, cnt) in sums.items():
                res.setdefault(k, {})[c] = tot / cnt
                res[k]["launches"] = cnt
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for k, d in res.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["fetch"], d["write"] = int(d["FETCH_SIZE"] * 1024 * 2), int(d["WRITE_SIZE"] * 1024)
            d["traffic"] = d["fetch"] + d["write"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE", 0) > 0:
            d["mfma_pipe_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    return (res, "") if res else (None, "not collected: the counter files held none of the kernels")


def stft_stage(pre, batches, launches: int = 210, warm: int = 60) -> dict:
    """The STFT stage on its own (cough_spectrogram = the reference's T.Spectrogram, preprocessing.py:131-136):
    waveform in, 257x101 power spectrogram out, timed with HIP events on the launch stream.  Reported beside the
    headline because BASELINE.json quotes an HBM fraction "for the STFT stage"; the fused featuriser above never
    writes this tensor."""
    import torch
    b = batches[0].shape[0]
    spec = torch.empty((b, 257, 101), dtype=torch.float32, device=batches[0].device)
    for i in range(warm):
        pre.spectrogram_batch(batches[i % len(batches)], out=spec)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(launches):                      # inputs rotate over distinct batches: every read comes from HBM
        pre.spectrogram_batch(batches[i % len(batches)], out=spec)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    achieved = b * BYTES_PER_CLIP_STFT / (ms * 1e-3) / 1e9
    traffic, src = None, None
    for name in ("r05_stft_pmc.json", "r04_stft_pmc.json", "r03_stft_pmc.json"):   # committed rocprofv3 PMC record of the same kernel, newest first
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                p = json.load(f)
            if p["clips_per_launch"] == b:
                traffic, src = int(p["traffic_bytes_per_launch"]), "profiles/" + name
                break
        except (OSError, KeyError, ValueError):
            pass
    return {"kernel": "stft3_kernel (waveform -> 257x101 power spectrogram; persistent, one 12-wave workgroup per CU)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "ms_per_launch": round(ms, 4),
            "algorithmic_bytes_per_launch": b * BYTES_PER_CLIP_STFT, "traffic": traffic, "traffic_source": src,
            "traffic_over_algorithmic": round(traffic / (b * BYTES_PER_CLIP_STFT), 3) if traffic else None}


def cpu_baseline(budget_s: float) -> dict:
    """Reference-faithful CPU path (per-clip loop, batch 1, STFT computed twice, softmax(...).item(),
    as src/preprocessing.py:398,425 + src/inference.py:216-217) and a best-effort batched CPU path,
    both from oracle/ (kind "port": the reference's own preprocessing.py needs torchaudio, absent here)."""
    from cough_detector_amd import synth
    from oracle import featurizer as ofeat, resnet as ores
    sd = synth.random_state_dict(seed=3)
    from cough_detector_amd.hostcpu import cpu_share
    default_threads = max(1, min(torch.get_num_threads(), cpu_share()))
    wav = torch.from_numpy(synth.make_clips(0, 256, peak_normalize=False))

    def faithful_leg(seconds):
        for i in range(4):                                                  # warm-up
            f = ofeat.extract_features(ofeat.normalize(wav[i:i + 1]))
            torch.softmax(ores.forward(f.unsqueeze(0), sd), dim=1)[0, 1].item()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            w = wav[n % 256:n % 256 + 1]
            f = ofeat.extract_features(ofeat.normalize(w))
            torch.softmax(ores.forward(f.unsqueeze(0), sd), dim=1)[0, 1].item()
            n += 1
        return n, n / (time.perf_counter() - t0)

    with torch.no_grad():
        # batch-1 ops are tiny: the default thread count of a many-core host only adds contention, so
        # the per-clip loop is timed at 1, 8 and the default number of threads and the best is reported
        best, n, threads = 0.0, 0, 1
        for th in sorted({1, min(8, default_threads), default_threads}):
            torch.set_num_threads(th)
            cnt, rate = faithful_leg(budget_s / 4)
            if rate > best:
                best, n, threads = rate, cnt, th
        faithful = best
        torch.set_num_threads(default_threads)
        ofeat.extract_features_batched_fast(wav, normalize_first=True)      # warm-up
        m, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / 4:
            f = ofeat.extract_features_batched_fast(wav, normalize_first=True)
            torch.softmax(ores.forward(f.unsqueeze(1), sd), dim=1)
            m += 256
        batched = m / (time.perf_counter() - t0)
    return {"value": round(faithful, 1), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{n} clips, per-clip loop (batch 1, duplicated STFT, softmax.item()) on synthetic 1 s clips; "
                      f"torch {torch.__version__} CPU, {threads} threads (cgroup share {cpu_share()} of {os.cpu_count()} logical CPUs)",
            "batched_value": round(batched, 1),
            "batched_sample": f"{m} clips at batch 256, single STFT, {default_threads} threads"}


def build_launch_cmd(argv, n_ranks: int, port: int):
    """The child command a ``--gpus N`` parent runs: torchrun's module entry with one rank per GPU of this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(args, argv) -> int:
    """Parent of a multi-rank run.  It stays subprocess-only and never touches the GPU runtime.  It refuses to start only
    on EXPLICIT evidence of a shortage -- a ``*_VISIBLE_DEVICES`` variable that names fewer devices than ``--gpus``
    (``hostcpu.explicit_device_limit``).  The sysfs count (``hostcpu.visible_gpu_count``: KFD topology + render-node
    access) is a heuristic that may under-count in an unfamiliar container, so a shortage there is only a warning: the
    ranks' own ``torch.cuda.device_count()`` decides, and a rank without a device says so and exits 2."""
    from cough_detector_amd.hostcpu import explicit_device_limit, visible_gpu_count
    limit = explicit_device_limit()
    if limit is not None and limit < args.gpus:
        print(f"bench.py --gpus {args.gpus}: only {limit} device(s) visible on this node "
              f"(ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES names {limit})", file=sys.stderr)
        return 2
    have = visible_gpu_count()
    if have is not None and have < args.gpus:
        print(f"bench.py --gpus {args.gpus}: WARNING: sysfs shows only {have} device(s) (KFD topology / render nodes); "
              f"launching anyway -- each rank checks its own device", file=sys.stderr)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    proc = subprocess.run(build_launch_cmd(argv, args.gpus, port), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    for ln in proc.stdout.splitlines():
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)                     # anything else the ranks wrote to stdout
    if lines:
        print(lines[-1], flush=True)
    if proc.returncode != 0:
        return proc.returncode
    return 0 if lines else 1


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096, help="clips per rank per step")
    ap.add_argument("--dtype", default=os.environ.get("COUGH_BENCH_DTYPE", "bf16x3"), choices=["bf16x3", "bf16_approx", "fp32"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--featurize-only", action="store_true", help="time K1 alone (BASELINE configs[1])")
    ap.add_argument("--rotate", type=int, default=3, help="distinct device-resident input batches the steps cycle over")
    ap.add_argument("--prewarm-s", type=float, default=0.6, help="untimed GPU pre-warm before --warmup (seconds)")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: steps per all-gather of logits (one bucketed exchange per this many steps)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not collect roofline.traffic with rocprofv3 --pmc child runs (N = 1); use the committed PMC record")
    ap.add_argument("--stft-launches", type=int, default=210,
                    help="timed launches of the stand-alone STFT stage (roofline_stft); the PMC child runs use a handful")
    ap.add_argument("--total-clips", type=int, default=0,
                    help="configs[3]: one pass over a stream of this many distinct clips (all ranks together); "
                         "overrides --steps and --rotate")
    args = ap.parse_args(argv)
    if args.dtype == "bf16":              # the pre-r03 name of the approximate mode (argparse does not check an env default)
        print("bench.py: COUGH_BENCH_DTYPE=bf16 is the APPROXIMATE single-bf16 mode, now called bf16_approx", file=sys.stderr)
        args.dtype = "bf16_approx"
    if args.dtype not in DTYPE_LABEL:
        ap.error(f"unknown dtype {args.dtype!r} (COUGH_BENCH_DTYPE / --dtype): choose from {sorted(DTYPE_LABEL)}")
    return args


def main():
    args = parse_args()
    force_dist = os.environ.get("COUGH_BENCH_FORCE_DIST") == "1"   # rehearse the multi-rank path with one rank
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        sys.exit(launch_ranks(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    from cough_detector_amd.hostcpu import bind_to_gpu_numa, bound_torch_threads
    share_gpu = os.environ.get("COUGH_BENCH_SHARE_GPU") == "1"       # REHEARSAL only: several ranks on one card
    backend = os.environ.get("COUGH_BENCH_BACKEND", "nccl")          # REHEARSAL only: "gloo" (RCCL refuses two ranks per GPU)
    if share_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    # multi-rank: pin this rank's host threads to the NUMA node of ITS GPU (sysfs only, before any GPU call and before
    # the thread pools exist); single process: leave the affinity alone
    numa = bind_to_gpu_numa(local_rank) if (world > 1 and os.environ.get("COUGH_BENCH_NUMA", "1") == "1") else \
        {"numa_node": None, "pci": None, "cpus": None}
    bound_torch_threads()          # size host thread pools to the cgroup CPU share (else the process is throttled)
    if torch.cuda.device_count() <= local_rank:       # a rank may touch the runtime; the launching parent may not
        print(f"bench.py --gpus {world}: only {torch.cuda.device_count()} device(s) visible on this node "
              f"(rank {rank} needs device {local_rank})", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and numa["pci"] is not None:
        # the sysfs order may not be the runtime's: check the PCI address of the device this rank really got
        from cough_detector_amd.hostcpu import rebind_to_pci_numa
        props = torch.cuda.get_device_properties(dev)
        if all(hasattr(props, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            numa = rebind_to_pci_numa(f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0", numa)
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL writes its version banner and warnings to stdout (at the first collective); stdout must carry exactly
        # one JSON line, so library output is routed to stderr until that line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        dist.barrier()        # builds the communicator now (tens of ms): the barriers around the timed region are then
                              # short, and the GPU does not idle its clocks down just before the first timed step

    import cough_detector_amd as cda
    from cough_detector_amd import synth
    from cough_detector_amd.distributed import BucketedLogitsGather, stream_shard

    B, W = args.batch, args.warmup
    # rank r owns global clips r, r+N, r+2N, ... (round-robin); clip g is generated on the device from seed g
    if args.total_clips > 0:
        batches, step_total = stream_shard(args.total_clips, B, rank, world, dev)
        K = len(batches)                                                     # steps of the rank with the most clips
    else:
        K = args.steps
        R = max(1, args.rotate)
        pool = torch.empty((R * B, 16000), dtype=torch.float32, device=dev)
        for j in range(R):
            synth.device_clips(rank + j * B * world, B, seed_stride=world, out=pool[j * B:(j + 1) * B])
        batches = [pool[(j % R) * B:(j % R + 1) * B] for j in range(K)]
        step_total = [B * world] * K
    pre = cda.AudioPreprocessor(device="cuda", **SHIPPED)
    model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype=args.dtype)
    model.load_state_dict(synth.random_state_dict(seed=3))
    model.to(dev).eval()
    feats = torch.empty((B, 90, 101), dtype=torch.float32, device=dev)
    pipe = cda.CoughPipeline(pre, model)
    fused = args.dtype in ("bf16_approx", "bf16x3") and not args.featurize_only   # the stem runs inside the featurise kernel
    # every rank ends up with every clip's logits: the steps' logits are exchanged in buckets of --gather-every steps,
    # one all-gather per bucket on RCCL's stream while the next bucket is computed
    publisher = BucketedLogitsGather(B, args.gather_every, dev) if dist else None

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]

    for e3 in ev:            # create the underlying HIP events (the library re-records them around K1)
        for e in e3:
            e.record()

    def step(i, timed):
        wav = batches[i % K]          # may be empty on the last step of a ragged --total-clips stream (still gathers)
        if args.featurize_only:
            if timed:
                ev[i][0].record()
            pre.featurize_batch(wav, normalize=True, out=feats[:wav.shape[0]])
            if timed:
                ev[i][1].record()
            return None
        # one C-ABI call: featurise (+ stem) -> residual blocks -> head; features are not materialised
        logits = pipe(wav, normalize=True, events=(ev[i][0], ev[i][1]) if timed else None)
        if timed:
            ev[i][2].record()
        if dist and not skip_gather:
            if i % K == 0:
                publisher.flush()                         # a bucket never spans the end of the stream (ragged last step)
            publisher.push(logits, step_total[i % K])     # every exchange is finished inside the timed region (drain)
            return publisher.out
        return logits

    skip_gather = os.environ.get("COUGH_BENCH_SKIP_GATHER") == "1"   # diagnostic only: ranks and barriers, no exchange

    def drain():
        if publisher is not None:
            publisher.drain()

    # ---- untimed pre-warm: the same pipeline until the wall clock says --prewarm-s (GPU at sustained clocks) ----
    t_pw = time.perf_counter()
    n_pw = 0
    while time.perf_counter() - t_pw < args.prewarm_s:
        for i in range(20):
            step((n_pw + i) % K, False)
        n_pw += 20
        drain()
        torch.cuda.synchronize()
    prewarm_s = time.perf_counter() - t_pw
    for i in range(W):
        step(i % K, False)
    drain()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i, True)
    drain()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel times are quoted on full batches; a --total-clips stream shorter than one batch has only a ragged step
    full = [i for i in range(K) if batches[i].shape[0] == B] or [i for i in range(K) if batches[i].shape[0] > 0]
    Bk = batches[full[0]].shape[0] if full else 0                  # clips per launch the kernel figures refer to
    k1_ms = sum(ev[i][0].elapsed_time(ev[i][1]) for i in full) / max(len(full), 1)
    net_ms = 0.0 if args.featurize_only else sum(ev[i][1].elapsed_time(ev[i][2]) for i in full) / max(len(full), 1)

    rank_records = None
    if dist:
        # what proves the collective path ran: every rank's own device / clock / kernel times / exchange counters,
        # gathered with ONE small all-gather after the timed region
        from cough_detector_amd.distributed import gather_rank_records
        rank_records = gather_rank_records({
            "rank": rank, "device": local_rank, "numa_node": -1 if numa["numa_node"] is None else numa["numa_node"],
            "host_cpus": -1 if numa["cpus"] is None else numa["cpus"],
            "ms_per_step": elapsed_local / max(K, 1) * 1e3, "k1_ms": k1_ms, "cls_ms": net_ms,
            "clips": sum(int(b_.shape[0]) for b_ in batches[:K]),
            "collectives_started": publisher.collectives if publisher else 0,
            "collectives_finished": publisher.finished if publisher else 0}, dev)
    if rank == 0:
        total_clips = sum(step_total)
        # the featurise stage's algorithmic bytes are 64 000 read + 36 360 written per clip, fused or not.
        # The fused kernel writes the stem's activation a1 instead of the feature image (bf16x3: 70 400 B of f32 per
        # clip); that hand-off shows up in `traffic` (PMC), not in `achieved`.
        k1_bytes = BYTES_PER_CLIP
        achieved = Bk * k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        pmc = measured_pmc(Bk, ("k1_fused_" + args.dtype) if fused else "k1")
        traffic, traffic_src = pmc.get("traffic_bytes_per_launch"), pmc.get("source")
        live_all = None
        if traffic is not None:
            traffic = int(traffic)
        if (world == 1 and not dist and not args.no_live_pmc and args.total_clips == 0 and Bk == B
                and os.environ.get("COUGH_BENCH_LIVE_PMC", "1") == "1"):
            live_all, why_not = live_pmc(args)
            if live_all is None and traffic_src:
                traffic_src = f"{traffic_src} (live PMC {why_not})"
            live = (live_all or {}).get("featurize_kernel")
            if live is not None and "traffic" in live:
                traffic = live["traffic"]
                traffic_src = (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two child runs of this command, {live['launches']} launches "
                               f"each): FETCH_SIZE[KiB]*1024*2 + WRITE_SIZE[KiB]*1024 = {live['fetch']} + {live['write']} bytes per launch")
        if args.featurize_only:
            workload = "configs[1]: batch=4096 synthetic 1s@16kHz mono -> 90x101 features, f32"
        elif args.total_clips > 0:
            workload = (f"configs[3]: {args.total_clips}-clip synthetic stream, every clip distinct, generated on-device "
                        f"from its index, resident in HBM, sharded round-robin, one pass in batches of {B} per rank -> "
                        "90x101 features -> CoughDetectorResidual logits, all-gather of logits")
        else:
            workload = ("configs[2]: batch=4096 synthetic 1s@16kHz mono per GPU -> 90x101 features "
                        "(64 mel + 13 MFCC + 13 delta, f32) -> CoughDetectorResidual logits")
        line = {
            "metric": "1s@16kHz clips/sec (featurise+infer)" if not args.featurize_only
                      else "1s@16kHz clips/sec (featurise only)",
            "value": round(total_clips / elapsed, 1), "unit": "clips/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if args.total_clips > 0 else "weak", "vs_baseline": None,
            "dtype": "f32" if args.featurize_only else DTYPE_LABEL[args.dtype], "data": "synthetic",
            "prewarm_s": round(prewarm_s, 3),
            "config": {"workload": workload, "clips_per_gpu_per_step": B,
                       "sharding": f"round-robin over {world} rank(s)" +
                                   (f" -- REHEARSAL: backend {backend}, ranks share one GPU: not a scaling measurement"
                                    if (share_gpu or backend != "nccl") else ""),
                       "collective": (f"all_gather(logits) once per {args.gather_every} steps ({args.gather_every * B * 8} B per rank), "
                                      "overlapped with the next bucket's compute") if dist else "none",
                       "inputs": (f"{len(batches)} distinct device-resident batches" if args.total_clips > 0 else
                                  f"{max(1, args.rotate)} distinct device-resident batches in rotation "
                                  f"({max(1, args.rotate) * B * 64000 / 2**20:.0f} MiB)") +
                                 ", generated on the device (cough_synth_clips)",
                       "weights": "random-init, BN stats randomised"},
            "roofline": {"kernel": f"featurize_kernel<stem fused, {args.dtype}> (K1+K2)" if fused else "featurize_kernel (K1)",
                         "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_algorithmic": round(traffic / (Bk * k1_bytes), 3) if traffic else None,
                         "ms_per_launch": round(k1_ms, 4),
                         "algorithmic_bytes_per_launch": Bk * k1_bytes, "clips_per_launch": Bk,
                         "bytes_moved_per_launch": Bk * (BYTES_PER_CLIP_FUSED[args.dtype] if fused else BYTES_PER_CLIP),
                         # the roof that actually binds K1: VALU issue (PMC SQ_ACTIVE_INST_VALU / SIMD quad-cycle slots)
                         "valu_issue_frac": pmc.get("valu_issue_frac"),
                         "valu_insts_per_clip": pmc.get("valu_insts_per_clip")},
        }
        if not args.featurize_only and net_ms > 0:
            tf = Bk * (FLOP_PER_CLIP_NO_STEM if fused else FLOP_PER_CLIP) / (net_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.dtype]
            line["roofline_classifier"] = {"kernel": "residual blocks + head (K3-K5)" + ("" if fused else " + stem (K2)"),
                                           "bound": "mfma",
                                           "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                                           "frac": round(tf / peak, 4), "ms_per_forward": round(net_ms, 4),
                                           "mfma_per_product": MFMA_PER_PRODUCT[args.dtype],
                                           "mfma_issue_frac": round(MFMA_PER_PRODUCT[args.dtype] * tf / peak, 4)}
            busy = {name: (live_all or {}).get(k, {}).get("mfma_pipe_busy")
                    for name, k in (("block0", "resblock_x3_kernel<32"), ("block1", "resblock_x3_kernel<64"))}
            if all(v is not None for v in busy.values()):
                # PMC, collected in this run: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) per kernel
                line["roofline_classifier"]["mfma_pipe_busy"] = busy
                line["roofline_classifier"]["mfma_pipe_busy_source"] = "live: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (child run of this command)"
        stft_in = [b_ for b_ in batches[:8] if b_.shape[0] == B]
        if world == 1 and not dist and stft_in:
            line["roofline_stft"] = stft_stage(pre, stft_in, launches=max(1, args.stft_launches),
                                               warm=min(60, max(1, args.stft_launches)))
            lv = (live_all or {}).get("stft3_kernel")
            if lv is not None and "traffic" in lv:     # the child runs launched the stand-alone STFT kernel too
                rs = line["roofline_stft"]
                rs["traffic"] = lv["traffic"]
                rs["traffic_source"] = (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE ({lv['launches']} launches): "
                                        f"{lv['fetch']} + {lv['write']} bytes per launch")
                rs["traffic_over_algorithmic"] = round(lv["traffic"] / rs["algorithmic_bytes_per_launch"], 3)
        if world == 1 and not dist and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        if dist:
            line["rccl_world"] = dist.get_world_size()
            line["backend"] = ("rccl (torch.distributed 'nccl' on ROCm)" if dist.get_backend() == "nccl"
                               else dist.get_backend())
            line["ranks"] = rank_records
            line["collectives"] = {"started": min(r["collectives_started"] for r in rank_records),
                                   "finished": min(r["collectives_finished"] for r in rank_records),
                                   "per": f"{args.gather_every} steps", "skipped": skip_gather}
            line["numa_binding"] = ({"pci": numa["pci"], "numa_node": numa["numa_node"], "cpus": numa["cpus"],
                                     "rebound_after_init": numa.get("rebound", False)} if world > 1 else None)
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        if dist:
            os.dup2(2, 1)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
