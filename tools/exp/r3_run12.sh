#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3l; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -8
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
for c in c3 c4 c6; do
for v in "" "SBEV_CHAIN_MAX_ROWS=1024"; do env $v python bench.py --config $c $Q --steps 30 2>/dev/null | tail -1 > $O/bench_$c.json; python -c "
import json
d=json.load(open('$O/bench_$c.json'))
print('$c [$v] value', d['value'], 'ms', d['ms_per_step'], 'launches', d['config']['launches_per_layer'])
"; done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o bench -- python $R/bench.py $Q --config c3 --steps 10 --warmup 3 > $O/kt_c3.log 2>&1
cp $(find $O/kt_c3 -name "*kernel_stats.csv" | head -1) $O/c3_kernel_stats.csv; rm -rf $O/kt_c3
grep "row_chain" $O/c3_kernel_stats.csv | cut -c1-200
