#!/bin/bash
# round 4, GPU call 19: tail chain on pairs of workgroups -- parity, then timings
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4_19
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -12
SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_chain_trace.so SBEV_NO_GRAPH=1 python bench.py --no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 3 --warmup 1 2>/dev/null | grep "^launch" | grep "PRE 0" | sed -n 8,12p | cut -c1-1200
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 30"
for c in c2 c5 c6 c1; do
python bench.py --config $c $Q 2>/dev/null | python tools/exp/bline.py "$c pairs   "
SBEV_NO_CHAIN_PAIR=1 python bench.py --config $c $Q 2>/dev/null | python tools/exp/bline.py "$c no pairs"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py $Q --steps 20 > $O/kt.log 2>&1
python $R/tools/exp/kstats.py $(find $O/kt -name "*kernel_stats.csv" | head -1) 10
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv")
