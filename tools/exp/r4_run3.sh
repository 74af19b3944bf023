#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_bf16s.py tests/test_gpu_fused.py -x -q -k "weight_stationary or nonfinite or generator" 2>&1 | tail -4
for v in "" exp_nch1 exp_nch4 exp_lock exp_nostore; do
  echo "== variant ${v:-product}"
  if [ -n "$v" ]; then export SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_$v.so; else unset SBEV_LIB_PATH; fi
  timeout 200 python tools/bench_gen_ws.py --shapes 900x32768 3200x32768 1600x77824 2>&1 | grep '^gen'
done
