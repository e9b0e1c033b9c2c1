#!/usr/bin/env python3
"""Aggregate the rocprofv3 --pmc CSVs that tools/pmc_kernel.sh left under <outdir> for kernels whose name contains <substr>."""
import collections, csv, glob, os, sys
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
