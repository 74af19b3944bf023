#!/bin/bash
# round 4, GPU call 24: launch order of the fused gather + mixing items (sbev_query_order): tests, then A/B at c2 / c5 / c6 / c3,
# raster and shuffled query rows; c2 with live PMC (fabric bytes of the fused launch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_order.py tests/test_gpu_fused.py tests/test_gpu_stepgraph.py -x -q 2>&1 | tail -5
Q="--no-cpu-baseline --no-alt --no-detector --steps 40"
for o in 0 1; do
python bench.py --config c2 --query-order $o $Q 2>/dev/null | tee -a gpurun_out/r4_run24_lines.jsonl | python tools/exp/bline.py "c2 order=$o        "
done
Q="$Q --no-live-pmc"
for o in 0 1; do
python bench.py --config c2 --query-order $o --shuffle-queries $Q 2>/dev/null | tee -a gpurun_out/r4_run24_lines.jsonl | python tools/exp/bline.py "c2 order=$o shuffled"
done
for c in c5 c6 c3; do
for o in 0 1; do
python bench.py --config $c --query-order $o --steps 20 --no-cpu-baseline --no-alt --no-detector --no-live-pmc 2>/dev/null | tee -a gpurun_out/r4_run24_lines.jsonl | python tools/exp/bline.py "$c order=$o        "
done
done
