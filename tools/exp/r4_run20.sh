#!/bin/bash
# round 4, GPU call 20: pair tail -- trace + timings after a change
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4_20
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q  2>&1 | tail -3

Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 30"
for c in c2 c5 c6 c3; do
python bench.py --config $c $Q 2>/dev/null | python tools/exp/bline.py "$c pairs   "
SBEV_NO_CHAIN_PAIR=1 python bench.py --config $c $Q 2>/dev/null | python tools/exp/bline.py "$c no pairs"
done
cd /tmp && export TMPDIR=/tmp
for c in c2 c3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -o bench -- python $R/bench.py --config $c $Q --steps 20 > $O/kt_$c.log 2>&1
python $R/tools/exp/kstats.py $(find $O/kt_$c -name "*kernel_stats.csv" | head -1) 12 | grep "row_chain\|sasa"
done
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv")
