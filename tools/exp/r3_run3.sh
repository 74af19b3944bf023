#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16s.py -x -q > $O/t_bf16s.log 2>&1; echo "bf16s tests rc=$?"; tail -5 $O/t_bf16s.log
echo "v2"; python tools/bench_bf16s.py --quick --m 900 3600 2>/dev/null
echo "v1"; SBEV_BF16S_GEN_V1=1 SBEV_BF16S_OUT_V1=1 python tools/bench_bf16s.py --quick --m 900 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-detector --no-live-pmc --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'host_issue', d['host_issue_ms_per_step'])
for k,v in d.get('alt_gemm',{}).items(): print(k, {a:b for a,b in v.items() if a!='gemm'})
"
