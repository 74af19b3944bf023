#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_bf16s.py -x -q 2>&1 | tail -8
python tools/exp/ablate_f16.py
bash tools/exp/r3_run16.sh
