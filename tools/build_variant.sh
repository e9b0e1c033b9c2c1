#!/bin/bash
# Build a variant of libcough_amd.so with extra -D flags into build_ab/ (same-box A/B timing via COUGH_AMD_LIB / tools/ab.sh).
# Usage: bash tools/build_variant.sh <name> [-DFOO=1 ...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build_ab
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function -fno-slp-vectorize -fvisibility=hidden -fvisibility-inlines-hidden -Wl,--version-script=cough_detector_amd/csrc/exports.map "$@" -o build_ab/lib_$NAME.so \
    cough_detector_amd/csrc/{api,featurize,featurize_generic,spectrogram,resnet,cnn,stream,synth}.hip
echo build_ab/lib_$NAME.so
