"""Dev tool: print one decoder step's kernel timeline (start / end in us relative to the step's first kernel) from a
rocprofv3 --kernel-trace CSV: which kernels really overlap?   python tools/exp/timeline.py <kernel_trace.csv> [step_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps start at the first transpose / linear3 kernel after a strip... simply: split on 'linear3_ln_relu' occurrences far apart
names = [r['Kernel_Name'] for r in rows]
starts = [i for i, n in enumerate(names) if 'transpose_tiles' in n and (i == 0 or 'transpose_tiles' not in names[i - 1])]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
i0, i1 = starts[k], starts[k + 1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:min(i1, i0 + 32)]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print('%8.1f %8.1f  %6.1f us  %s' % (s, e, e - s, r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]))
