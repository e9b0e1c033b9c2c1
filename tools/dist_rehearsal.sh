#!/bin/bash
# one-GPU rehearsal of everything multi-rank that CAN run on one card: 1 RCCL rank, 2 gloo ranks sharing the card, the --gpus 2 parent
set -e
OUT=gpurun_out/r05/dist
mkdir -p $OUT
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_bench_line.py tests/test_gpu_stream_job.py -m gpu -q -rs > $OUT/pytest_multirank.txt 2>&1
COUGH_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err
COUGH_CHECK_BACKEND=gloo COUGH_CHECK_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tests/multirank_check.py > $OUT/gloo_2rank_check.txt 2>&1
COUGH_BENCH_BACKEND=gloo COUGH_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err
python bench.py --gpus 2 > $OUT/bench_gpus2_parent.out 2> $OUT/bench_gpus2_parent.err || echo "parent rc=$?" >> $OUT/bench_gpus2_parent.err
tail -3 $OUT/pytest_multirank.txt; cat $OUT/bench_dist1.json; tail -4 $OUT/gloo_2rank_check.txt; cat $OUT/bench_gloo2.json; tail -5 $OUT/bench_gpus2_parent.err
