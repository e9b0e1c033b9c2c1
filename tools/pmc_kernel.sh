#!/bin/bash
# rocprofv3 PMC passes (separate passes; never combined with tracing) for one kernel of one command.
# Usage on the GPU box: bash tools/pmc_kernel.sh <outdir-under-gpurun_out> <kernel-name-substring> <script.py> [args...]
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/$1; KERN=$2; shift 2
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
export COUGH_BENCH_LIVE_PMC=0   # bench.py must not start rocprofv3 children of its own under this profiler
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$SCRIPT" "${ARGS[@]}" > "$OUT/$name.log" 2>&1
}
ARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run ea TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM
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
            print(f"{os.path.basename(d.rstrip('/')):6s} {k:28s} per-launch {v / max(n, 1):16.1f}  (launches {n})")
PY
