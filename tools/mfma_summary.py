"""Summarise a rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` pass (tools/profile_round.sh)
into profiles/<round>_mfma_summary.json: per kernel, MFMA-pipe utilisation = busy cycles summed over the SIMDs /
(GUI-active cycles x 1024 SIMDs) -- rocprofv3's own `MfmaUtil` expression.  GRBM_GUI_ACTIVE spans MORE than the traced kernel
(about 15 us of dispatch / counter start-stop around every serialised launch: the quotient GUI-active cycles / traced duration comes
out at 3.1-3.3 "GHz" for 20-us kernels on a 2.4-GHz part), so that expression is a LOWER bound and the quotient is NOT a clock --
round 4's `clock_ghz` field was this artefact.  With the in-kernel clocks of tools/gemm_clock.py
(profiles/r5_gemm_clock.json: s_memtime cycles / s_memrealtime wall time per workgroup) as a fourth argument the summary adds, for
the kernels stamped there, the utilisation inside the kernel: busy cycles / (traced duration x in-kernel clock x 1024 SIMDs).
Usage: python tools/mfma_summary.py <counter_csv> <kernel_trace_csv> <out_json> [<gemm_clock_json>]"""
import collections
import csv
import json
import sys

SIMDS = 256 * 4
XCDS = 8        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs of the device


def main():
    counters, trace, out = sys.argv[1:4]
    clocks = {}
    if len(sys.argv) > 4:
        try:
            for kname, rec in json.load(open(sys.argv[4]))['steady']['c2']['kernels'].items():
                clocks[kname.split(' ')[0]] = rec['shader_clock_ghz_median']
        except Exception as e:      # noqa: BLE001
            print('mfma_summary: no in-kernel clocks (%r)' % (e,), file=sys.stderr)
    dur = {}
    for r in csv.DictReader(open(trace)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    per = collections.defaultdict(dict)
    name = {}
    for r in csv.DictReader(open(counters)):
        per[r['Dispatch_Id']][r['Counter_Name']] = per[r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        name[r['Dispatch_Id']] = r['Kernel_Name']
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
    for d, c in per.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c or d not in dur:
            continue
        a = agg[name[d].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]]
        a[0] += c['SQ_VALU_MFMA_BUSY_CYCLES']; a[1] += c['GRBM_GUI_ACTIVE'] / XCDS; a[2] += dur[d]; a[3] += 1
    res = {}
    for k, (busy, gui, t, n) in agg.items():
        if busy <= 0 or n == 0:
            continue
        res[k] = {'launches': n, 'mfma_busy_cycles_per_launch': round(busy / n), 'gui_active_cycles_per_launch': round(gui / n),
                  'avg_us_under_pmc': round(1e6 * t / n, 2), 'mfma_util_pct': round(100 * busy / (gui * SIMDS), 1),
                  'gui_active_cycles_over_traced_duration_ghz_NOT_a_clock': round(gui / t / 1e9, 3) if t > 0 else None}
        f = next((v for kk, v in clocks.items() if kk in k), None)
        if f and t > 0:
            res[k]['clock_ghz_in_kernel'] = f
            res[k]['mfma_util_pct_in_kernel'] = round(100 * busy / (t * f * 1e9 * SIMDS), 1)
    json.dump({'note': 'mfma_util_pct = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE per XCD * 1024 SIMDs) = rocprofv3 MfmaUtil, a LOWER bound: '
                       'GUI_ACTIVE spans ~15 us more than the traced kernel; mfma_util_pct_in_kernel = busy / (traced duration x in-kernel clock of '
                       'tools/gemm_clock.py x 1024); counter passes serialise kernels, durations are NOT comparable with un-profiled runs', 'kernels': res},
              open(out, 'w'), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]['mfma_util_pct']):
        print('%-44s util >= %5.1f %%  in kernel %s %% at %s GHz  %8.1f us  (%d launches)' % (k[:44], v['mfma_util_pct'], v.get('mfma_util_pct_in_kernel'), v.get('clock_ghz_in_kernel'), v['avg_us_under_pmc'], v['launches']))


if __name__ == '__main__':
    main()
