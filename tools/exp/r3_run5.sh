#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SBEV_LIB_PATH=$R/sparsebev_amd/csrc/build/libsbev_exp_trace.so python tools/exp/trace_bf16s.py 2>/dev/null
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3e; mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -o b -- python $R/tools/bench_bf16s.py --quick --m 900 > $O/pmc.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('$O/pmc/**/*counter_collection.csv',recursive=True)
t=glob.glob('$O/pmc/**/*kernel_trace.csv',recursive=True)
dur={}
for r in csv.DictReader(open(t[0])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp'])-int(r['Start_Timestamp']))
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:58]
    if 'bf16s' not in k: continue
    acc[k].append(float(r['Counter_Value'])/dur[r['Dispatch_Id']])
for k,v in acc.items(): print(k, 'GRBM_GUI_ACTIVE/ns = clock GHz', round(sum(v)/len(v),3))
PY
rm -rf $O/pmc
