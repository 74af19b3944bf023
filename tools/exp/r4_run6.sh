#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python -m pytest tests/test_gpu_fused.py -x -q -k nonfinite 2>&1 | grep -v '^$' | tail -40
