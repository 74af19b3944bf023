"""Dev tool: top kernels of a rocprofv3 --kernel-trace --stats run (kernel_stats.csv): calls, average us, share."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in rows[:n]:
    name = r['Name']
    name = name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print('%-72s calls %6s  avg %9.2f us  %5.1f %%' % (name[:72], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
