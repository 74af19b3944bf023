#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_decoder.py tests/test_gpu_backward.py -x -q -k "attn or sasa or attention or g5 or g7 or dn_ or g6" 2>&1 | tail -4
SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_exp_sasa.so python tools/exp/r4_sasa_trace.py 2>&1 | tail -8
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 40"
python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2             "
python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2             "
