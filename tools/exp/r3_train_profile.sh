#!/bin/bash
# refresh the training-step files of a profile round only (see tools/profile_round.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${1:-r3}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_train.py > $OUT/train.log 2>&1; tail -1 $OUT/train.log
python $R/tools/bench_train.py --feat-grad --dropout >> $OUT/train.log 2>&1; tail -1 $OUT/train.log
rm -rf $OUT/kt_train
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_train -o train -- python $R/tools/bench_train.py --steps 5 > $OUT/kt_train.log 2>&1
rm -f $(find $OUT/kt_train -name "*kernel_trace.csv") $(find $OUT/kt_train -name "*agent_info.csv")
