#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
COUGH_AMD_LIB=$PWD/build_ab/lib_fused_notail.so python -m pytest tests/test_gpu_resnet.py -m gpu -q -k "reference_goldens and bf16x3" > $OUT/pytest_fused_notail.txt 2>&1; tail -2 $OUT/pytest_fused_notail.txt
bash tools/ab.sh $PWD/cough_detector_amd/libcough_amd.so $PWD/build_ab/lib_fused.so $PWD/build_ab/lib_fused_notail.so 2>&1 | tee $OUT/ab_fused2.txt
export COUGH_AMD_LIB=$PWD/build_ab/lib_fused.so; bash tools/pmc_kernel.sh r04/pmc_fused "resblock_x3_fused" bench.py --steps 6 --warmup 2 --cpu-seconds 0 --prewarm-s 0 > $OUT/pmc_fused.txt 2>&1; tail -30 $OUT/pmc_fused.txt
