#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4_12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -3
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 40"
for i in 1 2; do
python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2 rot         "
SBEV_CHAIN_NO_ROT=1 python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2 no rot      "
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py $Q --steps 20 > $O/kt.log 2>&1
python $R/tools/exp/kstats.py $(find $O/kt -name "*kernel_stats.csv" | head -1) 12 | grep -i 'chain\|sasa'
SBEV_CHAIN_NO_ROT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt0 -o bench -- python $R/bench.py $Q --steps 20 > $O/kt0.log 2>&1
python $R/tools/exp/kstats.py $(find $O/kt0 -name "*kernel_stats.csv" | head -1) 12 | grep -i 'chain\|sasa'
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv")
