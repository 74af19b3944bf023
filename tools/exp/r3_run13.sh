#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -3
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
for c in c3 c4; do
for v in "" "SBEV_CHAIN_TIE_SMALL=1"; do env $v python bench.py --config $c $Q --steps 30 2>/dev/null | tail -1 > $O/bench_$c.json; python -c "
import json
d=json.load(open('$O/bench_$c.json'))
print('$c [$v] value', d['value'], 'ms', d['ms_per_step'], 'launches', d['config']['launches_per_layer'])
"; done; done
SBEV_LIB_PATH=$R/tools/exp/libsbev_trace.so SBEV_NO_GRAPH=1 python bench.py --config c3 $Q --steps 3 --warmup 1 2>/dev/null | grep "^launch" > $O/trace_c3.txt
sed -n 14,16p $O/trace_c3.txt | cut -c1-700
