#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16s.py tests/test_gpu_workloads.py -x -q 2>&1 | tail -3
timeout 200 python tools/bench_gen_ws.py --shapes 900x32768 3200x32768 1600x77824 2>&1 | grep '^gen'
