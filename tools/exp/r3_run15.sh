#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3o; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_dense.py -x -q 2>&1 | tail -3
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
for c in c2 c4 c6 c2; do
python bench.py --config $c $Q --steps 40 2>/dev/null | tail -1 > $O/bench_$c.json; python -c "
import json
d=json.load(open('$O/bench_$c.json'))
print('$c value', d['value'], 'ms', d['ms_per_step'], 'launches', d['config']['launches_per_layer'], 'sampler', d['roofline']['avg_us'], d['roofline']['frac'], 'fused', d.get('roofline_fused',{}).get('avg_us'))
"; done
