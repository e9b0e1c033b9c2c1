#!/bin/bash
# r04 experiment: both residual blocks of a clip in one workgroup (RBX_FUSED=1) vs the two-kernel chain.
OUT=gpurun_out/r04; mkdir -p $OUT
COUGH_AMD_LIB=$PWD/build_ab/lib_fused.so python -m pytest tests/test_gpu_resnet.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py -m gpu -q -s > $OUT/pytest_fused.txt 2>&1
grep -E "passed|failed|bf16x3" $OUT/pytest_fused.txt | grep -v Warn | tail -12
bash tools/ab.sh $PWD/cough_detector_amd/libcough_amd.so $PWD/build_ab/lib_fused.so 2>&1 | tee $OUT/ab_fused.txt
