#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_chain_trace.so SBEV_NO_GRAPH=1 python bench.py --no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 3 --warmup 1 2>/dev/null | grep "^launch" | grep "PRE 0" | sed -n 9,10p | cut -c1-2500
