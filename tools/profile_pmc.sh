#!/bin/bash
# Per-config PMC passes (run on the GPU box through gpurun):  tools/profile_pmc.sh r2 c2 c3 c4 c5 c1
# Three SEPARATE rocprofv3 --pmc passes of the same bench command per config (FETCH_SIZE costs 3 of the 4 TCC slots,
# WRITE_SIZE 2 -- they cannot share a pass; the L2 hit / miss pair is the third), --kernel-trace only beside them.
# tools/pmc_summary.py turns them into profiles/<round>_pmc_<config>.json, which bench.py reads for roofline.traffic.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=$1; shift
cd /tmp && export TMPDIR=/tmp
for CFG in "$@"; do
  OUT=$R/gpurun_out/pmc_$CFG
  mkdir -p $OUT
  CMD="python $R/bench.py --config $CFG --no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 4 --warmup 2"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- $CMD > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- $CMD > $OUT/write.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -o bench -- $CMD > $OUT/tcc.log 2>&1
  F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
  W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
  T=$(find $OUT/tcc -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summary.py "$F" "$W" $R/gpurun_out/${ROUND}_pmc_$CFG.json "$T" "$CFG"
  # keep only the summaries: the raw counter CSVs are tens of MB
  rm -rf $OUT/fetch $OUT/write $OUT/tcc
done
