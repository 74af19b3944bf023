#!/bin/bash
# round 4, GPU call 23: fused gather + mixing vs two launches at the big configs (c6, c4, c3, c5)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 20"
for c in c6 c4 c3 c5; do
python bench.py --config $c $Q 2>/dev/null | python tools/exp/bline.py "$c fused  "
SBEV_NO_SAMPLE_MIX=1 python bench.py --config $c $Q 2>/dev/null | python tools/exp/bline.py "$c 2 launch"
done
