#!/bin/bash
# rocprofv3 PMC passes for the featurise kernel (separate passes; never combined with tracing).
# Usage on the GPU box: bash tools/pmc_k1.sh <outdir-under-gpurun_out>
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${1:-pmc_k1}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export COUGH_BENCH_LIVE_PMC=0   # bench.py must not start rocprofv3 children of its own under this profiler
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$REPO/bench.py" --featurize-only --steps 5 --warmup 2 --cpu-seconds 0 > "$OUT/$name.log" 2>&1
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*/"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "featurize" not in r["Kernel_Name"]:
                continue
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(agg.items()):
            print(f"{os.path.basename(d.rstrip('/')):6s} {k:28s} per-launch {v / max(n, 1):16.1f}  (launches {n})")
PY
