#!/bin/bash
# round 4, GPU call 27: the full GPU suite on the final build, then the fp16-storage bench lines (c2 with live PMC)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
OUT=$R/gpurun_out/prof_r4
mkdir -p $OUT
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
python bench.py --feat-dtype fp16 --no-cpu-baseline --no-alt --no-detector --steps 50 2>/dev/null | tail -1 > $OUT/bench_fp16.json; python tools/exp/bline.py "c2 fp16 NCHW" < $OUT/bench_fp16.json
python bench.py --feat-dtype fp16 --nhwc $Q --steps 50 2>/dev/null | tail -1 > $OUT/bench_fp16_nhwc.json; python tools/exp/bline.py "c2 fp16 NHWC" < $OUT/bench_fp16_nhwc.json
for c in c3 c4; do python bench.py --config $c --feat-dtype fp16 $Q --steps 30 2>/dev/null | tail -1 > $OUT/bench_${c}_fp16.json; python tools/exp/bline.py "$c fp16 NCHW" < $OUT/bench_${c}_fp16.json; done
python bench.py $Q --steps 50 2>/dev/null | python tools/exp/bline.py "c2 fp32 NCHW (same box)"
