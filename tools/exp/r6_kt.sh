#!/bin/bash
# kernel-trace stats of the bench step: lazy (default) and dense relayout.  usage: tools/exp/r6_kt.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-c2}; shift
OUT=$R/gpurun_out/r6_kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
for mode in lazy dense; do
  if [ $mode = dense ]; then export SBEV_NO_SPARSE_RELAYOUT=1; else unset SBEV_NO_SPARSE_RELAYOUT; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -o b -- python $R/bench.py $Q --steps 20 "$@" > $OUT/$mode.log 2>&1
  f=$(find $OUT/$mode -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/kernel_stats_$mode.csv
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:16]:
    print('%8.2f us x %5s = %6.2f %%  %s' % (float(r['AverageNs']) / 1e3, r['Calls'], float(r['Percentage']), r['Name'][:110]))
PY
  rm -rf $OUT/$mode
done
