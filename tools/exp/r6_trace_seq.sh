#!/bin/bash
# per-launch durations of one replayed step (kernel trace, no stats): tools/exp/r6_trace_seq.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-c2}; shift
OUT=$R/gpurun_out/r6_seq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o b -- python $R/bench.py $Q --steps 12 --warmup 4 "$@" > $OUT/run.log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/seq.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last complete step: from the last copy_indirect_kernel back one
idx = [i for i, r in enumerate(rows) if 'copy_indirect_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name).split('(')[0][:60]
    print('%9.2f  dur %8.2f  gap %6.2f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = e
print('step span us', (prev_end - t0) / 1e3)
PY
rm -rf $OUT/t
cat $OUT/seq.txt
