#!/bin/bash
# round 6: the per-unit counters of DESIGN 11.1 for the fused gather + mixing launch at c6 (1600 queries, 15 frames, 8 points,
# 5 bf16 levels: Pin = 120) -- is the c2 conclusion (no unit saturated, the sum is) the same where the launch is 48 % of the step?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6_c6
mkdir -p $O
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --config c6 $Q --steps 3 --warmup 2"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD" "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/cnt$i -o b -- $CMD > $O/cnt$i.log 2>&1
  grep -ciE "rocprofv3.*(error|invalid)" $O/cnt$i.log
done
K='adaptive_mixing_kernel<8, true, 5|adaptive_mixing_kernelILi8ELb1ELi5E'
python $R/tools/pmc_generic.py "$K" $O/r6_fused_counters_c6.json $(find $O/cnt* -name "*counter_collection.csv") | tee $O/fused_counters_c6.txt
# the launch's duration from one of the traces
f=$(find $O/cnt1 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee -a $O/fused_counters_c6.txt
import csv, sys
d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if 'adaptive_mixing_kernel' in r['Kernel_Name'] and ('<8, true, 5' in r['Kernel_Name'] or 'ILi8ELb1ELi5E' in r['Kernel_Name'])]
print('fused launch duration under the counter pass: n=%d avg %.1f us' % (len(d), sum(d) / max(len(d), 1)))
PY
rm -rf $O/cnt[0-9]*/ $O/cnt*.log
