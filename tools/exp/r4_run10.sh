#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4_10
for i in 1 2 3; do python tools/bench_train.py 2>&1 | tail -1 | tee -a gpurun_out/r4_10/train.log; done
python tools/bench_train.py --feat-grad --dropout 2>&1 | tail -1 | tee -a gpurun_out/r4_10/train.log
SBEV_NO_GEN_WS=1 python tools/bench_train.py 2>&1 | tail -1 | sed 's/^/tiled generator: /' | tee -a gpurun_out/r4_10/train.log
