#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
for c in c3 c4 c6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -o bench -- python $R/bench.py $Q --config $c --steps 10 --warmup 3 > $O/kt_$c.log 2>&1
  cp $(find $O/kt_$c -name "*kernel_stats.csv" | head -1) $O/${c}_kernel_stats.csv
done
rm -rf $O/kt_c3 $O/kt_c4 $O/kt_c6
ls $O
