#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 600 python bench.py --no-cpu-baseline --no-detector --no-live-pmc --steps 30 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json
d=json.load(open('$O/bench.json'))
print('value', d['value'], 'host_issue', d['host_issue_ms_per_step'])
for k,v in d.get('alt_gemm',{}).items(): print(k, {a:b for a,b in v.items() if a!='gemm'})
"
