#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_bf16s.py -x -q 2>&1 | tail -3
echo RF4; python tools/bench_bf16s.py --quick --m 900 3600 2>/dev/null
echo RF2; SBEV_BF16S_GEN_RF=2 python tools/bench_bf16s.py --quick --m 900 2>/dev/null
SBEV_LIB_PATH=$PWD/sparsebev_amd/csrc/build/libsbev_exp_trace.so python tools/exp/trace_bf16s.py 2>&1 | grep -A14 "group 0"
