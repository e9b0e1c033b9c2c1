#!/bin/bash
# rocprofv3 kernel-trace stats of one python script; prints the top kernels.  Usage: bash tools/prof_stats.sh <outdir> <script.py> [args...]
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SCRIPT=$REPO/$1; shift
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
export COUGH_BENCH_LIVE_PMC=0   # bench.py must not start rocprofv3 children of its own under this profiler
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o p -- python3 "$SCRIPT" "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n = r['Name'].replace('cough::(anonymous namespace)::', '')[:78]
    print(f"{n:78s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
