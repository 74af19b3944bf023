#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_stepgraph.py -x -q 2>&1 | tail -15
timeout 600 python bench.py --no-cpu-baseline --no-detector --no-live-pmc --steps 30 2>$O/bench.err | tail -1 > $O/bench.json; tail -3 $O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'host_issue', d['host_issue_ms_per_step'], d['config']['launches_per_layer'], d['config']['step_graph'])
print({k: d['roofline'][k] for k in ('frac','avg_us','launches')}, {k: d['roofline_fused'][k] for k in ('frac','avg_us','launches')})
"
SBEV_NO_GRAPH=1 python bench.py --no-cpu-baseline --no-detector --no-live-pmc --no-alt --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('no-graph value', d['value'], 'host_issue', d['host_issue_ms_per_step'])"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
