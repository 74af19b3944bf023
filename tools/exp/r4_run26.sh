#!/bin/bash
# round 4, GPU call 26: kernel stats of the batch configs (c3, c4) and the 1600-query config after the k-split
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in c3 c4 c6; do
  OUT=$R/gpurun_out/kt_$c
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $R/bench.py --config $c --no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 12 --warmup 3 > $OUT/log.txt 2>&1
  S=$(find $OUT -name "*kernel_stats.csv" | head -1)
  cp "$S" $R/gpurun_out/r4_kstats_$c.csv
  find $OUT -name "*.csv" -delete
  python - <<P
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/r4_kstats_$c.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('$c total kernel time %.1f ms' % (tot/1e6))
for r in rows[:12]:
    print('  %-84s %5s %9.1f us %5.1f%%' % (r['Name'][:84], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
P
done
