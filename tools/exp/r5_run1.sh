#!/bin/bash
# round 5, GPU call 1 (literal gpurun command: `gpurun --timeout 2700 -- bash tools/exp/r5_run1.sh`):
#   1 the GPU suite (new: pair-fault word, Tap under reentrant checkpoints, per-stream step graphs, finish_outputs, G13 free-running bound)
#   2 the GEMM clock artefact: in-kernel shader clock un-profiled / isolated / beneath rocprofv3 --pmc + sysfs sclk / power samples
#   3 PMC byte counters with the repaired summary: c2, c5, c6 (launch order) and c5, c6 in query order; kernel stats of both orders
#   4 query-order A/B timings at c5 / c6 (0 = launch order, 1 = sorted every layer, 2 = sorted once per step), three runs each
#   5 fused gather + mixing A/B: product (bound_ctrl broadcasts) vs packed fmas (exppk), sampler packed fmas (exppks), 4 / 3 / 2 workgroups per CU
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r5_1
mkdir -p $O
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
export PYTHONUNBUFFERED=1

echo "== 1 pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python -m pytest tests/test_gpu_decoder.py -m gpu -q -s -k "g7_decoder_teacher" 2>&1 | grep -E "free-running|passed|failed|Error" > $O/free_running.log; tail -8 $O/free_running.log

echo "== 2 gemm clock"
export SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_expwgt.so
timeout 400 python tools/gemm_clock.py --out $O/r5_gemm_clock.json > $O/gemm_clock.log 2>&1; tail -22 $O/gemm_clock.log
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/clk_pmc -o clk -- python $R/tools/gemm_clock.py --only steady --configs c2 --label "beneath rocprofv3 --pmc GRBM_GUI_ACTIVE" --out $O/r5_gemm_clock_under_pmc.json > $O/gemm_clock_pmc.log 2>&1); tail -4 $O/gemm_clock_pmc.log
rm -rf $O/clk_pmc
unset SBEV_LIB_PATH

echo "== 3 PMC (repaired summary) + kernel stats per order"
cd /tmp && export TMPDIR=/tmp
for spec in "c2 0" "c5 0" "c5 2" "c6 0" "c6 2"; do
  set -- $spec; CFG=$1; ORD=$2
  export SBEV_QUERY_ORDER=$ORD
  P=$O/pmc_${CFG}_o$ORD; mkdir -p $P
  CMD="python $R/bench.py --config $CFG $Q --steps 4 --warmup 2"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -o bench -- $CMD > $P/fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -o bench -- $CMD > $P/write.log 2>&1
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $P/tcc -o bench -- $CMD > $P/tcc.log 2>&1
  F=$(find $P/fetch -name "*counter_collection.csv" | head -1); W=$(find $P/write -name "*counter_collection.csv" | head -1); T=$(find $P/tcc -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summary.py "$F" "$W" $O/r5_pmc_${CFG}_order$ORD.json "$T" "$CFG" 2>&1 | grep -E "adaptive_mixing_kernel |row_chain|sasa|msmv|no short name"
  rm -rf $P/fetch $P/write $P/tcc
  if [ $CFG != c2 ]; then
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -o b -- python $R/bench.py --config $CFG $Q --steps 20 --warmup 3 > $P/kt.log 2>&1
    S=$(find $P/kt -name "*kernel_stats.csv" | head -1); cp "$S" $O/r5_kstats_${CFG}_order$ORD.csv; rm -rf $P/kt
    python $R/tools/exp/kstats.py $O/r5_kstats_${CFG}_order$ORD.csv 8
  fi
done
unset SBEV_QUERY_ORDER
cd $R

echo "== 4 query-order timings"
for c in c5 c6; do for o in 0 1 2; do for i in 1 2 3; do
  python bench.py --config $c --query-order $o $Q --steps 30 2>/dev/null | tee -a $O/order_lines_${c}_o$o.jsonl | python tools/exp/bline.py "$c order=$o run $i"
done; done; done

echo "== 5 fused-kernel variants at c2 (and c3)"
for i in 1 2; do
  python bench.py $Q --steps 50 2>/dev/null | tee -a $O/ab_base.jsonl | python tools/exp/bline.py "c2 product run $i"
  SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_exppk.so python bench.py $Q --steps 50 2>/dev/null | tee -a $O/ab_exppk.jsonl | python tools/exp/bline.py "c2 exppk run $i"
  SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_exppks.so python bench.py $Q --steps 50 2>/dev/null | tee -a $O/ab_exppks.jsonl | python tools/exp/bline.py "c2 exppks run $i"
done
for pad in 0 10240 25600; do
  SBEV_EXP_MIX_PAD=$pad SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_expocc.so python bench.py $Q --steps 50 2>/dev/null | tee -a $O/ab_occ_$pad.jsonl | python tools/exp/bline.py "c2 lds pad $pad"
done
python bench.py --config c3 $Q --steps 30 2>/dev/null | python tools/exp/bline.py "c3 product"
SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_exppk.so python bench.py --config c3 $Q --steps 30 2>/dev/null | python tools/exp/bline.py "c3 exppk"
echo "== done"
