#!/bin/bash
# A/B of variant libraries (tools/build_variant.sh): usage r4_ab.sh <config> <reps> <tag> [<tag> ...]   ("base" = the product library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
c=$1; reps=$2; shift 2
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 40"
for rep in $(seq $reps); do
for t in "$@"; do
  if [ $t = base ]; then unset SBEV_LIB_PATH; else export SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_$t.so; fi
  python bench.py --config $c $Q 2>/dev/null | python tools/exp/bline.py "$c $t"
done
done
