#!/bin/bash
# rocprofv3 PMC passes on L2 <-> fabric (EA) traffic for one kernel.  Usage: bash tools/pmc_tcc.sh <outdir> <kernel-substr> <script.py> [args...]
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
run ea1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
run ea2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
run ea3 TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum
run ea4 TCC_WRITE_REQ_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_REQ_sum
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
            print(f"{os.path.basename(d.rstrip('/')):6s} {k:40s} per-launch {v / max(n, 1):16.1f}  (launches {n})")
PY
