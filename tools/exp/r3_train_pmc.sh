#!/bin/bash
# HBM read bytes and LDS bank conflicts of the training step's kernels (separate rocprofv3 --pmc passes, --kernel-trace only beside them)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r3/pmc_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_train.py --steps 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o t -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -o t -- $CMD > $OUT/sq.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ('fetch', 'sq'):
    fs = glob.glob('$OUT/%s/**/*counter_collection.csv' % sub, recursive=True)
    if not fs:
        print(sub, 'no counter file'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        for key in ('gemm_tn_f16s', 'mixing_bwd', 'gemm_bf16s_gen3', 'gemm_bf16s_out3'):
            if key in k:
                acc[key][r['Counter_Name']] += float(r['Counter_Value']); n[key].add(r['Dispatch_Id'])
    for key in acc:
        print(sub, key, 'launches', len(n[key]), {c: round(v / len(n[key]), 1) for c, v in acc[key].items()})
PY
rm -rf $OUT/fetch $OUT/sq
