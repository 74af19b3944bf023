"""Per-kernel averages of arbitrary rocprofv3 --pmc counters: python tools/pmc_generic.py <kernel substring> <out_json> <counter_collection.csv>...
Each CSV is one pass (counters cannot always share a pass); values are summed over the dimensions rocprofv3 reports per dispatch
(XCDs, shader engines) and averaged over the dispatches whose kernel name contains the substring.  SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (4 shader cycles), SQ_VALU_MFMA_BUSY_CYCLES counts cycles
(/opt/skills/guides/MI355X_MICROARCH.md, latency table)."""
import collections
import csv
import json
import sys


def main():
    sub, out = sys.argv[1], sys.argv[2]
    res = {}
    for path in sys.argv[3:]:
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(path)):
            if any(x in r['Kernel_Name'] for x in sub.split('|')):      # (alternatives: a demangled and a mangled spelling of one kernel)
                per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        n = len(per)
        agg = collections.defaultdict(float)
        for d in per.values():
            for k, v in d.items():
                agg[k] += v
        for k, v in agg.items():
            res[k] = {'per_launch': round(v / max(n, 1), 1), 'launches': n}
    json.dump({'kernel_contains': sub, 'counters': res}, open(out, 'w'), indent=1)
    for k, v in sorted(res.items()):
        print('%-34s %16.1f  (%d launches)' % (k, v['per_launch'], v['launches']))


if __name__ == '__main__':
    main()
