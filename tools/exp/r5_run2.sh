#!/bin/bash
# round 5, GPU call 2 (`gpurun --timeout 3000 -- bash tools/exp/r5_run2.sh`):
#   1 the GPU suite again, whole (call 1 stopped at a wrong assertion in a NEW test)
#   2 what saturates in the fused gather + mixing launch (4 / 3 / 2 workgroups per CU cost 0 / 5.6 / 20 %: a throughput, not a latency bound):
#     SQ busy / wait / per-unit active-instruction counters of that kernel, TA busy; the counter list of this box for the record
#   3 packed fmas in the fused kernel at c3 again (one run in call 1 said -5 % on the launch)
#   4 the round's profile set (tools/profile_round.sh r5: bench lines, kernel stats, in-kernel GEMM clocks on the RIGHT card + zero-operand
#     launches, MFMA busy, training) and the PMC byte counters of c3 / c4
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r5_2
mkdir -p $O
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
export PYTHONUNBUFFERED=1

echo "== 1 pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.log; tail -5 $O/pytest.log

echo "== 2 counters of the fused launch"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TD|TCC|GRBM|SPI)_[A-Za-z0-9_]+" | sort -u > $O/counters_avail.txt; wc -l $O/counters_avail.txt
CMD="python $R/bench.py --config c2 $Q --steps 4 --warmup 2"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD" "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/cnt$i -o b -- $CMD > $O/cnt$i.log 2>&1
  grep -iE "error|invalid|not found|unable" $O/cnt$i.log | head -2
done
python $R/tools/pmc_generic.py "adaptive_mixing_kernel<2, true, 4" $O/r5_fused_counters_c2.json $(find $O/cnt* -name "*counter_collection.csv") | tee $O/fused_counters.txt
python $R/tools/pmc_generic.py "msmv_fwd_kernel" $O/r5_sampler_counters_c2.json $(find $O/cnt* -name "*counter_collection.csv") > $O/sampler_counters.txt
python $R/tools/pmc_generic.py "gemm_f16s_gen_ws_kernel" $O/r5_gen_counters_c2.json $(find $O/cnt* -name "*counter_collection.csv") > $O/gen_counters.txt
python $R/tools/pmc_generic.py "gemm_bf16s_out4_kernel" $O/r5_out_counters_c2.json $(find $O/cnt* -name "*counter_collection.csv") > $O/out_counters.txt
python $R/tools/pmc_generic.py "row_chain_kernel<0" $O/r5_tail_counters_c2.json $(find $O/cnt* -name "*counter_collection.csv") > $O/tail_counters.txt
rm -rf $O/cnt[0-9]*/
cd $R

echo "== 3 packed fmas at c3"
for i in 1 2; do
  python bench.py --config c3 $Q --steps 30 2>/dev/null | python tools/exp/bline.py "c3 product run $i"
  SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_exppk.so python bench.py --config c3 $Q --steps 30 2>/dev/null | python tools/exp/bline.py "c3 exppk run $i"
done

echo "== 4 profile round r5"
bash tools/profile_round.sh r5 2>&1 | cut -c1-260
bash tools/profile_pmc.sh r5 c3 c4 2>&1 | grep -E "adaptive_mixing_kernel |row_chain|msmv|transpose|gemm"
echo "== done"
