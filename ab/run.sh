#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for n in base w3 pf pfw3; do
  echo -n "$n fused: "; SBEV_LIB_PATH=$GRAFT_REPO_ROOT/ab/lib_$n.so python bench.py --no-cpu-baseline --no-alt --no-detector --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
echo -n "unfused: "; SBEV_NO_SAMPLE_MIX=1 python bench.py --no-cpu-baseline --no-alt --no-detector --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
