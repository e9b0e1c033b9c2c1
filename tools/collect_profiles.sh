#!/bin/bash
# Copy the summaries of a tools/final_profile.sh batch (gpurun_out/<tag>/) into profiles/ under this round's names.
# Usage: bash tools/collect_profiles.sh r05/final3 r05
set -eu
SRC=gpurun_out/$1; R=$2; P=profiles
cp $SRC/bench_default_20.json $P/${R}_bench_default_20.json
cp $SRC/bench_default_long.json $P/${R}_bench_default_long.json
cp $SRC/bench_featurize_only.json $P/${R}_bench_featurize_only.json
cp $SRC/bench_bf16_approx.json $P/${R}_bench_bf16_approx.json
cp $SRC/bench_fp32.json $P/${R}_bench_fp32.json
cp $SRC/bench_total_1M.json $P/${R}_bench_total_1M.json
cp $SRC/bench_dist1.json $P/${R}_bench_dist1_rehearsal.json
cp $SRC/rccl_check_1rank.txt $P/${R}_rccl_check_1rank.txt
cp $SRC/streaming_64.json $P/${R}_streaming_64streams.json
cp $SRC/prof_stats/p_kernel_stats.csv $P/${R}_final_kernel_stats.csv
cp $SRC/prof_stats_stft/p_kernel_stats.csv $P/${R}_stft_kernel_stats.csv
cat $SRC/pmc_block0.txt $SRC/pmc_block1.txt > $P/${R}_resblock_pmc_counters.txt
cp $SRC/pmc_k1_fused_x3_sq.txt $P/${R}_k1_fused_x3_sq_counters.txt
cp $SRC/k1_fused_x3_pmc.json $P/${R}_k1_fused_x3_pmc.json
cp $SRC/stft_pmc.json $P/${R}_stft_pmc.json
cp $SRC/pmc_stft.txt $P/${R}_stft_pmc_counters.txt
cp $SRC/bench_models.txt $P/${R}_bench_models.txt
cp $SRC/bench_heights.txt $P/${R}_heights.txt
cp $SRC/bench_flags.txt $P/${R}_bench_flags_final.txt
cp $SRC/bench_generic.txt $P/${R}_bench_generic_featurizer.txt
cp $SRC/bench_fullband.txt $P/${R}_bench_fullband_final.txt
echo collected
