#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_workloads.py -x -q -k "fused or c4" 2>&1 | tail -3
for v in "" "SBEV_NO_FUSE_L5F32=1"; do env $v python bench.py --config c4 --no-cpu-baseline --no-detector --no-live-pmc --no-alt --steps 30 2>/dev/null | tail -1 > $O/bench_c4.json; python -c "
import json
d=json.load(open('$O/bench_c4.json'))
print('c4 [$v] value', d['value'], 'sampler us', d['roofline']['avg_us'], 'fused', d.get('roofline_fused',{}).get('avg_us'), d['config']['launches_per_layer'])
"; done
