#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_stepgraph.py tests/test_gpu_head.py tests/test_gpu_bench.py -x -q 2>&1 | tail -30
