#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun):  tools/profile_round.sh r2
# bench line (default command), rocprofv3 kernel stats of the same workload (default f16x3 GEMM mode, exact f32 MFMA, bf16x6), the
# MFMA-busy counter pass, the other configs' bench lines, one training step's kernel stats.  PMC byte counters per config:
# tools/profile_pmc.sh.  Everything lands in gpurun_out/prof_<round>/; copy what is to be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${1:-r6}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
tail -1 $OUT/bench_line.json | cut -c1-400
Q="--no-cpu-baseline --no-alt --no-detector --no-live-pmc"
# kernel stats of the same workload: the default mode of the two mixing GEMMs (f16x3), the exact f32-MFMA kernels, bf16x6
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $R/bench.py $Q --steps 20 > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_f32 -o bench -- python $R/bench.py $Q --steps 20 --gemm f32 > $OUT/kt_f32.log 2>&1
# in-kernel shader clocks of the two mixing GEMMs + sysfs sclk / power samples (tools/gemm_clock.py; needs the -DSBEV_EXP_WGTIME variant library)
CLK=$OUT/${ROUND}_gemm_clock.json
if [ -f $R/sparsebev_amd/csrc/build/libsbev_expwgt.so ]; then
  (cd $R && SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_expwgt.so python tools/gemm_clock.py --out $CLK > $OUT/gemm_clock.log 2>&1; grep -E "GHz|sysfs" $OUT/gemm_clock.log | head -24)
  SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_expwgt.so rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/clk_pmc -o clk -- python $R/tools/gemm_clock.py --only steady --configs c2 --label "beneath rocprofv3 --pmc GRBM_GUI_ACTIVE" --out $OUT/${ROUND}_gemm_clock_under_pmc.json > $OUT/gemm_clock_pmc.log 2>&1
  rm -rf $OUT/clk_pmc
fi
# MFMA-busy counter pass (its own run, --kernel-trace only beside it): default mode, then the exact kernels
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o bench -- python $R/bench.py $Q --steps 5 --warmup 2 > $OUT/pmc_mfma.log 2>&1
python $R/tools/mfma_summary.py $(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_mfma -name "*kernel_trace.csv" | head -1) $OUT/${ROUND}_mfma_summary.json $CLK | head -8
rm -f $(find $OUT/pmc_mfma -name "*counter_collection.csv")
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_f32 -o bench -- python $R/bench.py $Q --steps 5 --warmup 2 --gemm f32 > $OUT/pmc_mfma_f32.log 2>&1
python $R/tools/mfma_summary.py $(find $OUT/pmc_mfma_f32 -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_mfma_f32 -name "*kernel_trace.csv" | head -1) $OUT/${ROUND}_mfma_summary_f32.json | head -6
rm -f $(find $OUT/pmc_mfma_f32 -name "*counter_collection.csv")
for c in c1 c3 c4 c5 c6; do python $R/bench.py --config $c $Q --steps 30 2>/dev/null | tail -1 > $OUT/bench_$c.json; cut -c1-220 $OUT/bench_$c.json; done
# round 6: the dense-relayout A/B of the NCHW configs (on-demand relayout is the default) and their kernel stats
for c in c2 c3 c4; do python $R/bench.py --config $c $Q --steps 30 --dense-relayout 2>/dev/null | tail -1 > $OUT/bench_${c}_dense.json; cut -c1-200 $OUT/bench_${c}_dense.json; done
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$c -o bench -- python $R/bench.py --config $c $Q --steps 20 > $OUT/kt_$c.log 2>&1
done
for c in c3 c4 c6; do python $R/bench.py --config $c $Q --steps 30 --gemm f32 2>/dev/null | tail -1 > $OUT/bench_${c}_f32.json; cut -c1-200 $OUT/bench_${c}_f32.json; done
python $R/bench.py --nhwc $Q --steps 50 2>/dev/null | tail -1 > $OUT/bench_nhwc.json; cut -c1-200 $OUT/bench_nhwc.json
python $R/bench.py --online $Q --steps 50 2>/dev/null | tail -1 > $OUT/bench_online.json; cut -c1-200 $OUT/bench_online.json
SBEV_NO_SAMPLE_MIX=1 python $R/bench.py $Q --steps 50 2>/dev/null | tail -1 > $OUT/bench_unfused.json; cut -c1-200 $OUT/bench_unfused.json
# fp16 feature storage (the reference's eval mode before its out_fp32 cast): NCHW lists through the 2-byte relayout; c2 with live PMC
python $R/bench.py --feat-dtype fp16 --no-cpu-baseline --no-alt --no-detector --steps 50 2>/dev/null | tail -1 > $OUT/bench_fp16.json; cut -c1-200 $OUT/bench_fp16.json
python $R/bench.py --feat-dtype fp16 --nhwc $Q --steps 50 2>/dev/null | tail -1 > $OUT/bench_fp16_nhwc.json; cut -c1-200 $OUT/bench_fp16_nhwc.json
for c in c3 c4; do python $R/bench.py --config $c --feat-dtype fp16 $Q --steps 30 2>/dev/null | tail -1 > $OUT/bench_${c}_fp16.json; cut -c1-200 $OUT/bench_${c}_fp16.json; done
for i in 1 2 3; do python $R/tools/bench_train.py --graph 2>&1 | tail -1 >> $OUT/train.log; done; tail -3 $OUT/train.log
python $R/tools/bench_train.py --graph --feat-grad --dropout >> $OUT/train.log 2>&1; tail -1 $OUT/train.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_train -o train -- python $R/tools/bench_train.py --steps 5 > $OUT/kt_train.log 2>&1
find $OUT -name "*kernel_stats.csv"
rm -f $(find $OUT -name "*kernel_trace.csv") $(find $OUT -name "*agent_info.csv")
