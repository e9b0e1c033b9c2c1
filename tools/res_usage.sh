#!/bin/bash
# VGPR / spill / occupancy summary of one translation unit: tools/res_usage.sh featurize.hip [extra flags]
cd /root/repo/cough_detector_amd/csrc
f=$1; shift
hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -fvisibility=hidden -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/res_usage.o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|VGPRs Spill|Occupancy|error|warning" | sed 's/.*remark: *//;s/ *\[-Rpass[^]]*\]//g' \
 | python3 -c '
import subprocess, sys
name, row = None, []
def flush():
    if name is None: return
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = d.split("(float")[0].split("(cough")[0].replace("cough::(anonymous namespace)::", "").replace("void ", "")
    print(f"{d[:70]:70s} " + "  ".join(row))
for line in sys.stdin:
    line = line.strip()
    if line.startswith("Function Name:"):
        flush(); name, row = line.split(":", 1)[1].strip(), []
    else:
        row.append(line)
flush()'
