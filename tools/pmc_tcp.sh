#!/bin/bash
# rocprofv3 PMC passes on the CU's vector-memory path (TA / TCP = L1 / TD) for one kernel: how busy the 64 B/clk L1 path
# is, how many of its requests go on to L2 and how long those take.  Separate passes, never combined with tracing.
# Usage: bash tools/pmc_tcp.sh <outdir-under-gpurun_out> <kernel-substr> <script.py> [args...]
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/$1; KERN=$2; shift 2
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SCRIPT=$REPO/$1; shift
ARGS=("$@")
cd /tmp && export TMPDIR=/tmp
export COUGH_BENCH_LIVE_PMC=0   # bench.py must not start rocprofv3 children of its own under this profiler
run() { name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$SCRIPT" "${ARGS[@]}" > "$OUT/$name.log" 2>&1; }
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp3 TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
# (TA_* / TD_* counters abort rocprofv3 on this image)
#run ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
#run td1 TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, os, sys, collections
out, kern = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(os.path.join(out, "*/"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if kern not in r["Kernel_Name"]:
                continue
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(agg.items()):
            print(f"{os.path.basename(d.rstrip('/')):6s} {k:40s} per-launch {v / max(n, 1):18.1f}  (launches {n})")
PY
