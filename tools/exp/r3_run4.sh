#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16s.py -x -q > $O/t_bf16s.log 2>&1; echo "bf16s tests rc=$?"; tail -5 $O/t_bf16s.log
python tools/bench_bf16s.py --quick --m 900 3600 2>/dev/null
echo gen2/out2; SBEV_BF16S_GEN_V2=1 SBEV_BF16S_OUT_V2=1 python tools/bench_bf16s.py --quick --m 900 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc -o b -- python $R/tools/bench_bf16s.py --quick --m 900 > $O/pmc.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('$O/pmc/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:60]
    if 'bf16s' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    if r['Counter_Name']=='SQ_WAVE_CYCLES': n[k]+=1
for k,v in acc.items():
    print(k, n[k], {c: round(x/n[k]) for c,x in v.items()})
PY
rm -rf $O/pmc
