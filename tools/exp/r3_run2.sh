#!/bin/bash
# round-3 call 2: ablations of the split-bf16 kernels (variant libraries built by tools/build_variant.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
echo "base"; python tools/bench_bf16s.py --quick --m 900 2>/dev/null
for v in exp_nostore exp_nomfma exp_noglds exp_hotw exp_hotx exp_nomfma_nostore; do
  echo $v; SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_$v.so python tools/bench_bf16s.py --quick --m 900 2>/dev/null
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- python $R/tools/bench_bf16s.py --quick --m 900 > $O/kt.log 2>&1
cat $(find $O/kt -name "*kernel_stats.csv" | head -1) | head -12
