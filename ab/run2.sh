#!/bin/bash
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 40"
for rep in 1 2; do
for c in c3 c4 c5; do
  echo -n "$c fused: "; python bench.py --config $c $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  echo -n "$c unfused: "; SBEV_NO_SAMPLE_MIX=1 python bench.py --config $c $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
done
