#!/bin/bash
# Copy the judged summaries of a profile round (tools/profile_round.sh + tools/profile_pmc.sh, merged back into gpurun_out/ by
# gpurun) into profiles/ under the round's prefix.  usage: tools/collect_profiles.sh r3
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
ROUND=${1:-r6}
S=$R/gpurun_out/prof_$ROUND
D=$R/profiles
mkdir -p $D
cp $S/kt/bench_kernel_stats.csv $D/${ROUND}_bench_kernel_stats.csv
[ -f $S/kt_f32/bench_kernel_stats.csv ] && cp $S/kt_f32/bench_kernel_stats.csv $D/${ROUND}_bench_f32_kernel_stats.csv
[ -f $S/kt6/bench_kernel_stats.csv ] && cp $S/kt6/bench_kernel_stats.csv $D/${ROUND}_bench_bf16x6_kernel_stats.csv
[ -f $S/kt3/bench_kernel_stats.csv ] && cp $S/kt3/bench_kernel_stats.csv $D/${ROUND}_bench_bf16x3s_kernel_stats.csv
for c in c3 c4; do [ -f $S/kt_$c/bench_kernel_stats.csv ] && cp $S/kt_$c/bench_kernel_stats.csv $D/${ROUND}_bench_kernel_stats_$c.csv; done
[ -f $S/kt_train/train_kernel_stats.csv ] && cp $S/kt_train/train_kernel_stats.csv $D/${ROUND}_train_step_kernel_stats.csv
cp $S/train.log $D/${ROUND}_train_step.log
tail -1 $S/bench_line.json > $D/${ROUND}_bench_line.json
for c in c1 c3 c4 c5 c6 c2_dense c3_dense c4_dense c3_f32 c4_f32 c6_f32 nhwc online unfused fp16 fp16_nhwc c3_fp16 c4_fp16; do [ -s $S/bench_$c.json ] && cp $S/bench_$c.json $D/${ROUND}_bench_line_$c.json; done
cp $S/${ROUND}_mfma_summary.json $D/ 2>/dev/null
cp $S/${ROUND}_gemm_clock.json $S/${ROUND}_gemm_clock_under_pmc.json $D/ 2>/dev/null
cp $S/${ROUND}_mfma_summary_bf16x6.json $D/ 2>/dev/null
cp $S/${ROUND}_mfma_summary_f32.json $D/ 2>/dev/null
for f in $R/gpurun_out/${ROUND}_pmc_*.json; do [ -f $f ] && cp $f $D/; done
ls -la $D | grep ${ROUND}_
