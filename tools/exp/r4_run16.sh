#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q 2>&1 | tail -3
for i in 1 2 3; do python tools/bench_train.py 2>&1 | tail -1; done
timeout 300 python tools/exp/train_graph.py 2>&1 | tail -4
