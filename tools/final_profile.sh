#!/bin/bash
# Round-end measurement batch on the GPU box: benches, rocprofv3 kernel stats, PMC passes.  Outputs under gpurun_out/<tag>/.
# Usage: bash tools/final_profile.sh <tag>
set -u
TAG=${1:-r05/final}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python bench.py --steps 20 --warmup 5 > $OUT/bench_default_20.json 2> $OUT/bench_default_20.err && \
python bench.py --cpu-seconds 0 > $OUT/bench_default_long.json 2> $OUT/bench_default_long.err && \
python bench.py --steps 200 --cpu-seconds 0 --featurize-only > $OUT/bench_featurize_only.json 2> $OUT/bench_featurize_only.err && \
python bench.py --steps 200 --cpu-seconds 0 --dtype bf16_approx > $OUT/bench_bf16_approx.json 2> $OUT/bench_bf16.err && \
python bench.py --steps 50 --cpu-seconds 0 --dtype fp32 > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err && \
python bench.py --total-clips 1000000 --cpu-seconds 0 > $OUT/bench_total_1M.json 2> $OUT/bench_total_1M.err && \
COUGH_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err && \
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tests/multirank_check.py > $OUT/rccl_check_1rank.txt 2>&1 && \
python bench_streaming.py --streams 64 --seconds 20 > $OUT/streaming_64.json 2> $OUT/streaming_64.err && \
bash tools/prof_stats.sh $TAG/prof_stats bench.py --steps 200 --warmup 10 --cpu-seconds 0 > $OUT/prof_stats.txt 2>&1 && \
bash tools/prof_stats.sh $TAG/prof_stats_stft tools/bench_stft.py --launches 200 --rounds 1 > $OUT/prof_stats_stft.txt 2>&1 && \
bash tools/pmc_kernel.sh $TAG/pmc_all "resblock_x3_kernel<32" bench.py --steps 6 --warmup 2 --cpu-seconds 0 --prewarm-s 0 > $OUT/pmc_block0.txt 2>&1 && \
python tools/pmc_agg.py $OUT/pmc_all "resblock_x3_kernel<64" > $OUT/pmc_block1.txt && \
python tools/pmc_agg.py $OUT/pmc_all "featurize_kernel" > $OUT/pmc_k1_fused_x3_sq.txt && \
python tools/pmc_to_json.py $OUT/pmc_all $OUT/k1_fused_x3_pmc.json 4096 featurize 100360 > $OUT/k1_fused_x3_pmc.txt && \
bash tools/pmc_kernel.sh $TAG/pmc_stft "stft3_kernel" tools/bench_stft.py --launches 6 --rounds 1 --prewarm-s 0 > $OUT/pmc_stft.txt 2>&1 && \
python tools/pmc_to_json.py $OUT/pmc_stft $OUT/stft_pmc.json 4096 stft3 167828 > $OUT/stft_pmc.txt && \
python tools/bench_models.py > $OUT/bench_models.txt 2>&1 && python tools/bench_models.py --dtypes bf16x3 --iters 20 >> $OUT/bench_models.txt 2>&1 && \
python tools/bench_heights.py > $OUT/bench_heights.txt 2>&1 && \
python tools/bench_flags.py > $OUT/bench_flags.txt 2>&1 && python tools/bench_generic.py > $OUT/bench_generic.txt 2>&1 && \
python tools/bench_fullband.py > $OUT/bench_fullband.txt 2>&1 && \
echo FINAL_PROFILE_OK
