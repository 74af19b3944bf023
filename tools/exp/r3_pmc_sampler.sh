#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in c5 c2; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_$C -o b -- env SBEV_NO_SAMPLE_MIX=1 SBEV_NO_GRAPH=1 python $R/bench.py --config $C --no-cpu-baseline --no-alt --no-detector --no-live-pmc --steps 3 --warmup 2 > $O/pmc_$C.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('$O/pmc_$C/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:70]
    if 'msmv_fwd' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': n[k]+=1
for k,v in acc.items():
    print('$C', k, n[k], {c: round(x/n[k]) for c,x in v.items()})
PY
rm -rf $O/pmc_$C
done
