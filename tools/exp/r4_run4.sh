#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_bf16s.py -x -q -k "weight_stationary or generator" 2>&1 | tail -3
for v in "" exp_lock; do
  echo "== variant ${v:-product}"
  if [ -n "$v" ]; then export SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_$v.so; else unset SBEV_LIB_PATH; fi
  timeout 200 python tools/bench_gen_ws.py --shapes 900x32768 3200x32768 1600x77824 2>&1 | grep '^gen'
done
export SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_exp_trace.so
echo "=== trace M=900"; timeout 120 python tools/exp/r4_trace_ws.py 900 | sed -n '1,8p;17,24p;33,34p'
