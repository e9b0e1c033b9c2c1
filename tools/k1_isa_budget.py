#!/usr/bin/env python3
"""Per-phase instruction budget of the featurise kernel (K1) from its gfx950 ISA.

    python tools/k1_isa_budget.py [--kernel plain|bf16|bf16x3] > profiles/r02_k1_isa_budget.txt

Compiles csrc/featurize.hip with -DCOUGH_K1_MARKERS -save-temps (the markers are ISA comments between two scheduling
barriers: no instruction is added, but the scheduler cannot move instructions across a phase boundary, so the counts
are attributable -- the production build differs from this one only in instruction ORDER), cuts the chosen instantiation's instruction stream at the markers and counts wave-instructions per phase and
class.  Dynamic counts = static counts x the trip count of the enclosing loop: the P1 loop runs 26 four-frame groups
over 4 waves = 6.5 iterations per wave; everything the compiler left as a loop inside a phase is reported with its
static size and back-edge so that the trip count can be read from the source.  Sections between SKIP / ENDSKIP markers
(branches that the shipped configuration never takes) are left out.  Per clip = per wave x 4 waves.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"plain": "featurize_kernelILb0ELi0E", "bf16": "featurize_kernelILb0ELi1E", "bf16x3": "featurize_kernelILb0ELi2E"}

# trip counts per wave of loops the compiler keeps rolled, by phase (from the source): (static back-edge count order)
ROLLED_TRIPS = {
    "P2 floor + mel rows store": 3232 / 256.0,          # i2 < 64*101/2 step 256
    "P2 DCT 13x64": 64 / 8.0,                            # #pragma unroll 8 over the 64 mel bands
    "P2 feature image (stem) + MFCC / delta rows": None,  # several loops: reported individually
    "K2 stem MFMA + pool + store": 8.0,                  # 16 tiles, two per iteration
}


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("ds_", "s_barrier")):
        return "lds" if op.startswith("ds_") else "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="bf16x3", choices=list(KERNELS))
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", "-fno-slp-vectorize", "-DCOUGH_K1_MARKERS", "-c", "-save-temps",
               os.path.join(ROOT, "cough_detector_amd", "csrc", "featurize.hip"), "-o", os.devnull]
        subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(os.path.join(tmp, "featurize-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN5cough") and KERNELS[args.kernel] in l.split(":")[0] and ": ; @" in l)
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    phases, order = {}, []
    cur, mult, skip, weight = "entry", 1.0, False, 1.0
    labels, loops = {}, []
    for ln in range(start + 1, end + 1):
        l = text[ln].strip()
        if l.startswith("; K1MARK"):
            body = l[len("; K1MARK"):].strip()
            if body.startswith("PHASE"):
                cur = body[6:].strip()
            elif body.startswith("LOOP"):
                mult = float(body.split()[1])
            elif body.startswith("ENDLOOP"):
                mult = 1.0
            elif body.startswith("SKIP"):
                skip = True
            elif body.startswith("ENDSKIP"):
                skip = False
            elif body.startswith("WEIGHT"):
                weight = float(body.split()[1])
            elif body.startswith("ENDWEIGHT"):
                weight = 1.0
            continue
        if not l or l.startswith((";", ".")) and not l.startswith(".LBB"):
            continue
        if l.startswith(".LBB"):
            labels[l.split(":")[0]] = (ln, cur)
            continue
        op = l.split()[0]
        if skip:
            continue
        key = cur if mult == 1.0 else cur
        if key not in phases:
            phases[key] = {"mult": mult, "static": {}}
            order.append(key)
        c = classify(op)
        phases[key]["static"][c] = phases[key]["static"].get(c, 0) + weight
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt][0] < ln and mult == 1.0:       # back-edge outside the marked P1 loop
                n = sum(1 for k in range(labels[tgt][0], ln) if text[k].strip() and text[k].strip().split()[0].startswith("v_"))
                loops.append((cur, tgt, n))
    print(f"# K1 instruction budget, kernel = {args.kernel} ({KERNELS[args.kernel]}), gfx950, hipcc -O3")
    print("# wave-instructions PER WAVE (x4 waves = per clip); dynamic = static x trips of the enclosing marked loop")
    print(f"{'phase':62s} {'trips':>6s} {'VALU':>7s} {'MFMA':>6s} {'LDS':>6s} {'VMEM':>6s} {'SALU':>6s} {'SMEM':>5s} {'wait':>5s}")
    tot = {}
    for k in order:
        st, mu = phases[k]["static"], phases[k]["mult"]
        row = {c: st.get(c, 0) * mu for c in ("valu", "mfma", "lds", "vmem", "salu", "smem", "wait")}
        for c, v in row.items():
            tot[c] = tot.get(c, 0) + v
        print(f"{k:62s} {mu:6.1f} {row['valu']:7.0f} {row['mfma']:6.0f} {row['lds']:6.0f} {row['vmem']:6.0f} "
              f"{row['salu']:6.0f} {row['smem']:5.0f} {row['wait']:5.0f}")
    print(f"{'TOTAL per wave (rolled loops counted ONCE, see below)':62s} {'':6s} {tot['valu']:7.0f} {tot['mfma']:6.0f} "
          f"{tot['lds']:6.0f} {tot['vmem']:6.0f} {tot['salu']:6.0f} {tot['smem']:5.0f} {tot['wait']:5.0f}")
    print("# loops the compiler kept rolled outside the P1 loop (static VALU in the loop body; multiply by trips - 1 and add):")
    extra = 0.0
    for cur, tgt, n in loops:
        trips = ROLLED_TRIPS.get(cur)
        add = n * (trips - 1) if trips else 0.0
        extra += add
        print(f"#   phase '{cur}': loop at {tgt}: {n} VALU per iteration" + (f", {trips:.1f} trips -> +{add:.0f}" if trips else ", trips: see source"))
    print(f"# VALU per wave incl. rolled loops with known trips ~ {tot['valu'] + extra:.0f}; per clip (4 waves) ~ {4 * (tot['valu'] + extra):.0f}")


if __name__ == "__main__":
    main()
