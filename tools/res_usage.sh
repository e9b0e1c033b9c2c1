#!/bin/bash
# VGPR / spill / occupancy summary of one translation unit: tools/res_usage.sh featurize.hip [extra flags]
cd /root/repo/cough_detector_amd/csrc
f=$1; shift
hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -fvisibility=hidden -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/res_usage.o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|VGPRs Spill|Occupancy|error|warning" | sed 's/.*remark: *//;s/\[-Rpass[^]]*\]//g;s/Function Name: //' | paste - - - - - \
 | awk -F'\t' '{cmd="c++filt " $1; cmd | getline d; close(cmd); sub(/\(.*/, "", d); sub(/.*::/, "", d); print d "\t" $2 $3 $4 $5}'
