#!/bin/bash
# Dev tool: build an A/B variant of libsbev_hip.so with extra -D flags on ONE translation unit, next to the real
# library (sparsebev_amd/csrc/build/libsbev_<tag>.so); select it at run time with SBEV_LIB_PATH.
# usage: tools/build_variant.sh <tag> <unit.hip> <flags...>
set -e
cd "$(dirname "$0")/../sparsebev_amd/csrc"
tag=$1; unit=$2; shift 2
python -m sparsebev_amd.csrc.build >/dev/null 2>&1 || (cd ../.. && python -m sparsebev_amd.csrc.build >/dev/null)
extra=""
case $unit in msmv_sampling_bwd.hip) extra="-munsafe-fp-atomics";; head.hip|project.hip|mixing.hip|backward_ops.hip) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" -c $unit -o build/${unit%.hip}_$tag.o
objs=""
for o in build/*.o; do
  b=$(basename $o .o)
  case $b in *_v[0-9]*|*_exp*) continue;; esac
  if [ "$b" = "${unit%.hip}" ]; then objs="$objs build/${unit%.hip}_$tag.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsbev_$tag.so $objs
echo build/libsbev_$tag.so
