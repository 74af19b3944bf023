#!/bin/bash
# round 4, GPU call 1: buffer-load sampler + weight-stationary generator -- correctness, then A/B timings; chain RG=2 experiment
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4_1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_fused.py -x -q 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_bf16s.py -x -q -k "weight_stationary or generator" 2>&1 | tail -6
timeout 300 python tools/bench_gen_ws.py 2>&1 | tail -8
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 30"
python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2 default     "
SBEV_NO_GEN_WS=1 python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2 tiled gen   "
SBEV_MSMV_NO_BUF=1 python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2 no-buf smplr"
SBEV_CHAIN_RG=2 python bench.py $Q 2>/dev/null | python tools/exp/bline.py "c2 chain RG=2  "
for c in c5 c6 c3; do
python bench.py --config $c $Q 2>/dev/null | python tools/exp/bline.py "$c default     "
SBEV_MSMV_NO_BUF=1 python bench.py --config $c $Q 2>/dev/null | python tools/exp/bline.py "$c no-buf smplr"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py $Q --steps 20 > $O/kt.log 2>&1
python $R/tools/exp/kstats.py $(find $O/kt -name "*kernel_stats.csv" | head -1) 14
SBEV_CHAIN_RG=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_rg2 -o bench -- python $R/bench.py $Q --steps 20 > $O/kt_rg2.log 2>&1
python $R/tools/exp/kstats.py $(find $O/kt_rg2 -name "*kernel_stats.csv" | head -1) 10
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv")
