#!/bin/bash
# round 4, GPU call 22: pair tail -- split of the in-projection between the members (A/B), then the full GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4_22
mkdir -p $O
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 20"
cd /tmp && export TMPDIR=/tmp
for q0 in 5 6 7 8; do
SBEV_CHAIN_PAIR_Q0=$q0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$q0 -o bench -- python $R/bench.py $Q > $O/kt_$q0.log 2>&1
echo "q0=$q0 $(python $R/tools/exp/kstats.py $(find $O/kt_$q0 -name "*kernel_stats.csv" | head -1) 12 | grep "row_chain_kernel<0")"
done
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv")
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
