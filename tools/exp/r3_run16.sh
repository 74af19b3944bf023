#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3p; mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
for g in f16x3 f16x4 bf16x6 f32; do
python bench.py --config c2 $Q --steps 40 --gemm $g 2>$O/err_$g.txt | tail -1 > $O/bench_c2_$g.json; python -c "
import json
d=json.load(open('$O/bench_c2_$g.json'))
print('c2 $g value', d['value'], 'ms', d['ms_per_step'], [ (k['kernel'][:40], k['avg_us']) for k in d.get('roofline_mfma',[])])
" || tail -5 $O/err_$g.txt; done
timeout 1500 python -m pytest tests/test_gpu_workloads.py -x -q -k "c2" 2>&1 | tail -5
