#!/bin/bash
# round-3 call 1: new split-bf16 kernels -- correctness, microbench, bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16s.py -x -q -s > $O/t_bf16s.log 2>&1; echo "bf16s tests rc=$?"; tail -15 $O/t_bf16s.log
timeout 300 python tools/bench_bf16s.py > $O/bench_bf16s.log 2>&1; cat $O/bench_bf16s.log
timeout 600 python bench.py --no-cpu-baseline --no-detector --no-live-pmc --steps 30 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
timeout 900 python -m pytest tests/test_gpu_workloads.py -x -q -k "c2" > $O/t_work.log 2>&1; echo "workload tests rc=$?"; tail -8 $O/t_work.log
