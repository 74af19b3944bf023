#!/bin/bash
# round 4, GPU call 18: feasibility of a PAIR split of the tail chain -- time the tail when every workgroup streams half of the weights
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4_18
mkdir -p $O
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 20"
cd /tmp && export TMPDIR=/tmp
for v in base rg2 half; do
  E=""
  [ $v = rg2 ] && E="SBEV_CHAIN_RG=2"
  [ $v = half ] && E="SBEV_CHAIN_RG=2 SBEV_CHAIN_EXP_HALF=1"
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o bench -- python $R/bench.py $Q > $O/kt_$v.log 2>&1
  echo "== $v"; python $R/tools/exp/kstats.py $(find $O/kt_$v -name "*kernel_stats.csv" | head -1) 8 | grep row_chain
done
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv")
