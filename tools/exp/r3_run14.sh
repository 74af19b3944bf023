#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_dense.py tests/test_gpu_decoder.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_workloads.py -x -q -k "c6 or c4" 2>&1 | tail -3
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
for c in c6 c2 c5; do
python bench.py --config $c $Q --steps 30 2>/dev/null | tail -1 > $O/bench_$c.json; python -c "
import json
d=json.load(open('$O/bench_$c.json'))
print('$c value', d['value'], 'ms', d['ms_per_step'], 'launches', d['config']['launches_per_layer'], 'sampler', d['roofline']['avg_us'], d['roofline']['frac'], 'fused', d.get('roofline_fused',{}).get('avg_us'))
"; done
SBEV_NO_SAMPLE_MIX=1 python bench.py --config c6 $Q --steps 30 2>/dev/null | tail -1 > $O/bench_c6_unfused.json; python -c "
import json
d=json.load(open('$O/bench_c6_unfused.json'))
print('c6 unfused value', d['value'], 'ms', d['ms_per_step'])"
python bench.py --config c6 $Q --steps 30 --gemm bf16x6 2>/dev/null | tail -1 > $O/bench_c6_x6.json; python -c "
import json
d=json.load(open('$O/bench_c6_x6.json'))
print('c6 bf16x6 value', d['value'], 'ms', d['ms_per_step'])"
