#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_bf16s.py -x -q 2>&1 | tail -3
python tools/bench_bf16s.py --quick --m 900 3600 2>/dev/null
SBEV_LIB_PATH=$PWD/sparsebev_amd/csrc/build/libsbev_exp_trace.so python tools/exp/trace_bf16s.py 2>&1 | grep -A8 "group 0"
SBEV_LIB_PATH=$PWD/sparsebev_amd/csrc/build/libsbev_exp_trace.so python tools/exp/trace_bf16s.py 2>&1 | tail -4
SBEV_LIB_PATH=$PWD/sparsebev_amd/csrc/build/libsbev_exp_trace.so python tools/exp/trace_out_bf16s.py 2>&1 | grep -A7 "half 0"
