#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the same command,
# and the two separate PMC passes (FETCH_SIZE / WRITE_SIZE) that tools/pmc_summary.py turns into bytes per launch.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
tail -1 $OUT/bench_line.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $R/bench.py --no-cpu-baseline --no-alt --steps 20 > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -o bench -- python $R/bench.py --no-cpu-baseline --no-alt --steps 20 --gemm bf16x3 > $OUT/kt3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py --no-cpu-baseline --no-alt --steps 5 --warmup 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py --no-cpu-baseline --no-alt --steps 5 --warmup 2 > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
for c in c1 c3 c4; do python $R/bench.py --config $c --no-cpu-baseline --no-alt --steps 30 2>/dev/null | tail -1 | cut -c1-200; done
python $R/bench.py --nhwc --no-cpu-baseline --no-alt --steps 50 2>/dev/null | tail -1 | cut -c1-200
python $R/bench.py --online --no-cpu-baseline --no-alt --steps 50 2>/dev/null | tail -1 | cut -c1-200
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o bench -- python $R/bench.py --no-cpu-baseline --no-alt --steps 5 --warmup 2 > $OUT/pmc_mfma.log 2>&1
