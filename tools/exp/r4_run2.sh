#!/bin/bash
# round 4, GPU call 2: ablations of the weight-stationary generator (no stores / no MFMAs / no LDS-DMA / neither), prefetch depth
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_bf16s.py tests/test_gpu_fused.py -x -q -k "weight_stationary or nonfinite" 2>&1 | tail -4
for v in "" exp_nostore exp_nomfma exp_noglds exp_nosm exp_pf2 exp_pf3; do
  echo "== variant ${v:-product}"
  if [ -n "$v" ]; then export SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_$v.so; else unset SBEV_LIB_PATH; fi
  timeout 200 python tools/bench_gen_ws.py --shapes 900x32768 3200x32768 2>&1 | grep '^gen'
done
