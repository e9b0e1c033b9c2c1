#!/bin/bash
# rocprofv3 PMC passes for any kernel of bench.py (separate passes; never combined with tracing).
# Usage on the GPU box: bash tools/pmc.sh <outdir-under-gpurun_out> <kernel-name-substring> [bench args...]
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/$1; KSUB=$2; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export COUGH_BENCH_LIVE_PMC=0   # bench.py must not start rocprofv3 children of its own under this profiler
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$REPO/bench.py" --steps 4 --warmup 2 --cpu-seconds 0 $BARGS > "$OUT/$name.log" 2>&1
}
BARGS="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES
run mem FETCH_SIZE
run memw WRITE_SIZE
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, os, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(os.path.join(out, "*/"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(f)):
            if ksub not in r["Kernel_Name"]:
                continue
            kn = r["Kernel_Name"].split("(")[0][-60:]
            a = agg[kn][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for kn, cs in agg.items():
            for k, (v, n) in sorted(cs.items()):
                print(f"{kn:60s} {k:28s} per-launch {v / max(n, 1):16.1f}  (n={n})")
PY
