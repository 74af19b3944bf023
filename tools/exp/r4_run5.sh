#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16s.py tests/test_gpu_sampling.py tests/test_gpu_fused.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_decoder.py tests/test_gpu_stepgraph.py -x -q 2>&1 | tail -3
