#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_stepgraph.py -x -q 2>&1 | tail -3
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 40"
for i in 1 2; do
python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2 fork        "
SBEV_NO_RELAYOUT_FORK=1 python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2 no fork     "
done
python bench.py --config c3 $Q 2>/dev/null | python tools/exp/bline.py "c3 fork        "
SBEV_NO_RELAYOUT_FORK=1 python bench.py --config c3 $Q 2>/dev/null | python tools/exp/bline.py "c3 no fork     "
python bench.py --config c4 $Q 2>/dev/null | python tools/exp/bline.py "c4 fork        "
SBEV_NO_RELAYOUT_FORK=1 python bench.py --config c4 $Q 2>/dev/null | python tools/exp/bline.py "c4 no fork     "
