"""Turn the rocprofv3 --pmc output of tools/pmc_k1.sh into profiles/<name>.json with the gfx950
corrections of MI355X_MICROARCH.md "HBM" applied (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reads 1/2
of the bytes of a coalesced stream -- confirmed for 4/8/16 B per lane by tools/calib_fetch.hip;
WRITE_SIZE is exact).  Usage: python tools/pmc_to_json.py gpurun_out/<dir> profiles/<name>.json [batch] [kernel-substring] [bytes/clip]"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
kern = sys.argv[4] if len(sys.argv) > 4 else "featurize"
bytes_per_clip = int(sys.argv[5]) if len(sys.argv) > 5 else 100360
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
per = {k: v / n for k, (v, n) in agg.items()}
fetch_b = per["FETCH_SIZE"] * 1024 * 2
write_b = per["WRITE_SIZE"] * 1024
out = {"kernel": kern + "_kernel", "clips_per_launch": batch, "normalize": True,
       "counters_per_launch": per,
       "corrections": "FETCH_SIZE[KiB]*1024*2 (gfx950 half-count, calibrated), WRITE_SIZE[KiB]*1024",
       "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
       "traffic_bytes_per_launch": fetch_b + write_b,
       "algorithmic_bytes_per_launch": batch * bytes_per_clip,
       "traffic_over_algorithmic": (fetch_b + write_b) / (batch * bytes_per_clip)}
# VALU-issue share, when the SQ pass is in the same directory: SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD
# issues VALU work; the launch offers GRBM_GUI_ACTIVE / 8 XCDs cycles on each of 1024 SIMDs = that many / 4 quad-cycle slots
if "SQ_ACTIVE_INST_VALU" in per and "GRBM_GUI_ACTIVE" in per:
    slots = per["GRBM_GUI_ACTIVE"] / 8.0 * 1024 / 4.0
    out["valu_issue_frac"] = round(per["SQ_ACTIVE_INST_VALU"] / slots, 4)
if "SQ_INSTS_VALU" in per:
    out["valu_insts_per_clip"] = round(per["SQ_INSTS_VALU"] / batch, 1)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "traffic_over_algorithmic")}))
