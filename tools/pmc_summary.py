"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into profiles/<round>_pmc_summary.json.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide (16 B/lane) coalesced reads, so it is doubled.
That factor is re-validated here on our own known-byte-count kernel (transpose_tiles_kernel reads every feature
byte exactly once with 16-B loads).  An optional third pass (--pmc TCC_HIT_sum TCC_MISS_sum) adds the L2 hit ratio
TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) per kernel (MI355X_MICROARCH.md, L2 section).
Usage: python tools/pmc_summary.py <fetch_csv> <write_csv> <out_json> [<tcc_csv> [<config name>]]"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def short(name):
    import re
    m = re.search(r'adaptive_mixing_kernel<\d+, (true|false), (\d+)', name)
    if m:                      # the fused gather + mixing instantiations (L > 0) vs the plain mixing kernel
        return 'adaptive_mixing_kernel' if m.group(2) != '0' else 'adaptive_mixing_kernel_plain'
    m = re.search(r'adaptive_mixing_kernelILi\d+ELb[01]ELi(\d+)E', name)      # (rocprofv3 leaves _Float16 instantiations mangled)
    if m:
        return 'adaptive_mixing_kernel' if m.group(1) != '0' else 'adaptive_mixing_kernel_plain'
    # the three row chains (csrc/row_chain.hip): first template argument 0 tail (+ next front), 1 layer-0 front, 2 attention chain.  ANY
    # number of further arguments (round 4's pair tail is row_chain_kernel<0, 2, true>: the two-argument pattern of round 3 dropped every
    # chain from profiles/r4_pmc_*.json without a word), demangled or mangled
    m = re.search(r'row_chain_kernel<(\d)\s*[,>]', name) or re.search(r'row_chain_kernelILi(\d)E', name)
    if m:
        return 'row_chain_kernel_' + {'0': 'tail', '1': 'front', '2': 'attention'}.get(m.group(1), m.group(1))
    for key in ('msmv_fwd_kernel', 'adaptive_mixing_kernel', 'transpose_tiles_multi_kernel', 'transpose_tiles_kernel', 'transpose_tiles16_kernel',
                'lazy_tiles_kernel', 'lazy_scan_kernel', 'finish_outputs_kernel', 'copy_indirect_kernel', 'query_order_kernel', 'sasa_kernel', 'splitk_reduce_kernel',
                'gemm_nt_f32_small_kernel', 'gemm_group_small_kernel', 'gemm_nt_f32_strip_kernel', 'gemm_nt_f32_regtile_kernel',
                'sample_project_kernel', 'sampling_front_kernel', 'ffn_fused_kernel', 'branch_chain_kernel', 'gemm_nt_f32_kernel<true', 'gemm_nt_f32_kernel<false', 'gemm_bf16x3',
                'gemm_f16s_gen_ws_kernel', 'gemm_bf16s_gen3_kernel', 'gemm_bf16s_out3_kernel', 'gemm_bf16s_out4_kernel', 'gemm_bf16s_out8_kernel', 'pack_frags_kernel'):
        if key in name:
            return key
    return None


def main():
    fetch, write, out = sys.argv[1:4]
    tcc = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] else None
    config = sys.argv[5] if len(sys.argv) > 5 else 'c2'
    f, w = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
    hit, miss = (per_kernel(tcc, 'TCC_HIT_sum'), per_kernel(tcc, 'TCC_MISS_sum')) if tcc else ({}, {})
    res = {}
    dropped = []
    for name in set(f) | set(w):
        s = short(name)
        if s is None:
            if 'anonymous namespace' in name or '_GLOBAL__N_' in name:      # one of this library's kernels without a short name: say so
                dropped.append(name.split('(')[0][:100])
            continue
        fk, n = f.get(name, (0.0, 0))
        wk, _ = w.get(name, (0.0, 0))
        prev = res.get(s)
        if prev is not None and prev['launches_sampled'] >= n:      # several instantiations share a short name: keep the most launched
            continue
        res[s] = {'launches_sampled': n, 'FETCH_SIZE_KiB_raw': round(fk, 1), 'WRITE_SIZE_KiB': round(wk, 1),
                  'fetch_bytes_corrected_x2': int(2 * fk * 1024), 'write_bytes': int(wk * 1024),
                  'hbm_bytes_per_launch': int(2 * fk * 1024 + wk * 1024)}
        if name in hit and name in miss and hit[name][0] + miss[name][0] > 0:
            h, m = hit[name][0], miss[name][0]
            res[s].update({'TCC_HIT_sum': round(h), 'TCC_MISS_sum': round(m), 'l2_hit_ratio': round(h / (h + m), 4)})
    json.dump({'config': config,
               'note': 'rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc TCC_HIT_sum TCC_MISS_sum in separate passes of `python bench.py --config %s '
                       '--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-detector` (tools/profile_pmc.sh); KiB units; FETCH_SIZE x2 (gfx950 wide-read '
                       'under-count, validated on transpose_tiles_kernel whose true read bytes are known); FETCH/WRITE are fabric-side L2 requests, i.e. '
                       'HBM + Infinity-Cache traffic' % config, 'kernels': res}, open(out, 'w'), indent=1)
    for name in sorted(set(dropped)):
        print('pmc_summary: no short name for library kernel %s (not in the summary)' % name, file=sys.stderr)
    for k, v in sorted(res.items()):
        print('%-28s fetch(corr) %8.1f MB  write %8.1f MB  L2 hit %s' % (k, v['fetch_bytes_corrected_x2'] / 1e6, v['write_bytes'] / 1e6, v.get('l2_hit_ratio')))


if __name__ == '__main__':
    main()
