#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py $Q --config c2 --steps 20 --warmup 3 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/c2_f16x3_kernel_stats.csv; rm -rf $O/kt
head -6 $O/c2_f16x3_kernel_stats.csv | cut -c1-150
tail -1 $O/kt.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d.get('gemm_gate'))"
