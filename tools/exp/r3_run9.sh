#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
for c in c3 c4 c5; do timeout 900 python bench.py --config $c --no-cpu-baseline --no-detector --no-live-pmc --steps 20 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; tail -2 $O/bench_$c.err; python -c "
import json
d=json.load(open('$O/bench_$c.json'))
print('$c value', d['value'], 'ms', d['ms_per_step'], d['config']['launches_per_layer'])
print(d['roofline']['avg_us'], d['roofline']['frac'], d.get('roofline_mfma'))
for k,v in d.get('alt_gemm',{}).items(): print(k, {a:b for a,b in v.items() if a!='gemm'})
"; done
true 2>&1 | tail -6
