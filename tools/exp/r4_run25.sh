#!/bin/bash
# round 4, GPU call 25: why the ordered launch is not faster -- fabric bytes / L2 hits of the fused kernel with and without the
# order (PMC passes), kernel durations (trace), and the sort in-stream (mode 2: every layer, mode 3: layer 0 only) vs the side stream
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 40"
for o in 0 1 2 3; do
python bench.py --config c2 --query-order $o $Q 2>/dev/null | python tools/exp/bline.py "c2 order=$o"
done
for o in 0 2 3; do
python bench.py --config c2 --shuffle-queries --query-order $o $Q 2>/dev/null | python tools/exp/bline.py "c2 shuffled order=$o"
done
cd /tmp && export TMPDIR=/tmp
for o in 0 2; do
  export SBEV_QUERY_ORDER=$o
  OUT=$R/gpurun_out/pmc_o$o
  mkdir -p $OUT
  CMD="python $R/bench.py --config c2 --no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 4 --warmup 2"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- $CMD > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- $CMD > $OUT/write.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -o bench -- $CMD > $OUT/tcc.log 2>&1
  F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
  W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
  T=$(find $OUT/tcc -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summary.py "$F" "$W" $R/gpurun_out/r4_pmc_c2_order$o.json "$T" "c2"
  rm -rf $OUT/fetch $OUT/write $OUT/tcc
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o b -- python $R/bench.py --config c2 --no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 20 --warmup 3 > $OUT/kt.log 2>&1
  S=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
  cp "$S" $R/gpurun_out/r4_kstats_c2_order$o.csv
  rm -rf $OUT/kt
done
cd $R
python - <<'P'
import json, csv
for o in (0, 2):
    d = json.load(open('gpurun_out/r4_pmc_c2_order%d.json' % o))['kernels']
    for k in ('adaptive_mixing_kernel', 'msmv_fwd_kernel'):
        if k in d: print('order', o, k, {x: d[k][x] for x in ('fetch_bytes_corrected_x2', 'write_bytes', 'hbm_bytes_per_launch', 'l2_hit_ratio')})
    for r in csv.DictReader(open('gpurun_out/r4_kstats_c2_order%d.csv' % o)):
        if any(t in r['Name'] for t in ('adaptive_mixing_kernel<2, true, 4', 'query_order', 'sasa_kernel', 'gen_ws', 'out4', 'row_chain')):
            print('   ', r['Name'][:70], r['Calls'], round(float(r['AverageNs']) / 1e3, 2))
P
