#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes for the fused featurise+stem kernel as bench.py runs it (separate passes).
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${1:-pmc_k1_fused}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
export COUGH_BENCH_LIVE_PMC=0   # bench.py must not start rocprofv3 children of its own under this profiler
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- \
      python3 "$REPO/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 > "$OUT/$c.log" 2>&1
done
