"""What clock do the two f16x3 mixing GEMMs actually run at?  (the "power-capped, 1.1-1.7 GHz" sentence of DESIGN 10.2 had no
artefact, and profiles/r4_mfma_summary.json -- a PMC pass -- shows the same kernels at 2.2-2.4 GHz.)

Needs a variant library whose gemm_bf16s.hip was built with -DSBEV_EXP_WGTIME (tools/build_variant.sh expwgt gemm_bf16s.hip -DSBEV_EXP_WGTIME;
SBEV_LIB_PATH=sparsebev_amd/csrc/build/libsbev_expwgt.so): every workgroup of the generator / out-projection kernels stamps wall clock
(s_memrealtime, 100 MHz) and shader clock (s_memtime) at its first and last instruction; cycles / wall time = the clock it ran at.
Three conditions, same stamps:
  steady    the kernels inside the decoder step, steps replayed back to back (what bench.py times)
  isolated  the same two launches alone (c2 / c3 operand shapes), device idle for a few ms before each -- what a serialising
            profiler pass sees
  (under rocprofv3 --pmc: run this tool with --only steady beneath the profiler; the run script does)
Beside them a sampler thread reads the amdgpu hwmon sysfs files (gfx clock, socket power) every few ms through an idle / busy / idle
window -- the driver's own view, at <= 10 ms.
Writes one JSON (default gpurun_out/r5_gemm_clock.json) and prints a summary.  Run from the repo root on the GPU box."""
import argparse
import ctypes
import glob
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebev_amd import _lib, dense, runtime, synthetic as S  # noqa: E402

KINDS = {0: 'gemm_bf16s_gen3_kernel (tiled generator)', 1: 'gemm_f16s_gen_ws_kernel (weight-stationary generator)',
         2: 'gemm_bf16s_out3_kernel (out-projection, 256-row tiles)', 3: 'gemm_bf16s_out4_kernel (out-projection, 128-row tiles)'}


def raw_lib():
    raw = ctypes.CDLL(_lib.LIB_PATH)
    if not hasattr(raw, 'sbev_debug_wgtime_read'):
        raise SystemExit('gemm_clock: %s has no workgroup stamps -- build gemm_bf16s.hip with -DSBEV_EXP_WGTIME and point SBEV_LIB_PATH at it' % _lib.LIB_PATH)
    raw.sbev_debug_wgtime_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return raw


def read_kind(raw, kind):
    buf = (ctypes.c_ulonglong * (1024 * 4))()
    if raw.sbev_debug_wgtime_read(buf, kind) != 0:
        return None
    a = np.array(buf, dtype=np.uint64).reshape(1024, 4).astype(np.int64)
    a = a[(a[:, 0] > 0) & (a[:, 2] > a[:, 0])]
    if len(a) == 0:
        return None
    us = (a[:, 2] - a[:, 0]) / 100.0                      # s_memrealtime ticks at 100 MHz
    cyc = (a[:, 3] - a[:, 1]).astype(np.float64)
    ghz = cyc / us / 1e3
    return {'workgroups': int(len(a)), 'launch_span_us': round(float((a[:, 2].max() - a[:, 0].min()) / 100.0), 2),
            'lifetime_us_median': round(float(np.median(us)), 2), 'lifetime_us_min': round(float(us.min()), 2), 'lifetime_us_max': round(float(us.max()), 2),
            'shader_clock_ghz_median': round(float(np.median(ghz)), 3), 'shader_clock_ghz_p10': round(float(np.percentile(ghz, 10)), 3),
            'shader_clock_ghz_p90': round(float(np.percentile(ghz, 90)), 3)}


class Sysfs(threading.Thread):
    """amdgpu hwmon: freq1_input (gfx clock, Hz), power1_average / power1_input (socket power, microwatt)"""

    def __init__(self, period=0.004):
        super().__init__(daemon=True)
        self.period, self.samples, self.stop_flag, self.marks = period, [], False, []
        self.files = {}
        # the box shows every GPU of the node in sysfs, this process owns ONE: pick the card whose PCI address is the HIP device's
        self.bdf = None
        try:
            pr = torch.cuda.get_device_properties(0)
            self.bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:      # noqa: BLE001
            pass
        cards = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'))
        mine = [hw for hw in cards if self.bdf and os.path.realpath(os.path.join(hw, '..', '..')).endswith(self.bdf)]
        self.card_matched = bool(mine)
        for hw in (mine or cards):
            for name, key in (('freq1_input', 'sclk_hz'), ('power1_average', 'power_uw'), ('power1_input', 'power_uw'), ('freq2_input', 'mclk_hz')):
                p = os.path.join(hw, name)
                if key not in self.files and os.path.exists(p):
                    try:
                        int(open(p).read().strip())
                        self.files[key] = p
                    except (OSError, ValueError):
                        pass
            if self.files:
                break
        self.listing = {hw: sorted(os.listdir(hw)) for hw in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')}

    def run(self):
        fds = {k: open(p) for k, p in self.files.items()}
        while not self.stop_flag:
            row = {'t': time.perf_counter()}
            for k, f in fds.items():
                try:
                    f.seek(0)
                    row[k] = int(f.read().strip())
                except (OSError, ValueError):
                    row[k] = None
            self.samples.append(row)
            time.sleep(self.period)

    def mark(self, name):
        self.marks.append((name, time.perf_counter()))

    def summary(self):
        out = {'files': self.files, 'hip_device_pci': self.bdf, 'card_matched_by_pci_address': self.card_matched, 'period_ms_actual': None, 'windows': {}}
        if len(self.samples) > 1:
            out['period_ms_actual'] = round(1e3 * (self.samples[-1]['t'] - self.samples[0]['t']) / (len(self.samples) - 1), 2)
        for (name, t0), (_, t1) in zip(self.marks[:-1], self.marks[1:]):
            rows = [r for r in self.samples if t0 <= r['t'] < t1]
            w = {'samples': len(rows), 'seconds': round(t1 - t0, 3)}
            for key, unit, div in (('sclk_hz', 'sclk_mhz', 1e6), ('power_uw', 'power_w', 1e6), ('mclk_hz', 'mclk_mhz', 1e6)):
                v = [r[key] / div for r in rows if r.get(key) is not None]
                if v:
                    w[unit] = {'min': round(min(v), 1), 'median': round(float(np.median(v)), 1), 'max': round(max(v), 1)}
            out['windows'][name] = w
            if name in ('idle_before_c2', 'busy_steady_c2'):      # the samples themselves, ms since the window opened
                out['series_' + name] = {'columns': ['t_ms', 'sclk_mhz', 'power_w'],
                                         'rows': [[round(1e3 * (r['t'] - t0), 1), (r.get('sclk_hz') or 0) / 1e6, (r.get('power_uw') or 0) / 1e6] for r in rows]}
        return out


def smi_snapshot():
    import subprocess
    for cmd in (['rocm-smi', '--showclocks', '--showpower', '--showperflevel', '--json'], ['amd-smi', 'metric', '-c', '-p', '--json']):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            if r.returncode == 0 and r.stdout.strip():
                return {'cmd': ' '.join(cmd), 'out': json.loads(r.stdout)}
        except Exception:      # noqa: BLE001
            pass
    return None


def decoder_step(config):
    import bench
    cfg = bench.CONFIGS[config]
    pyr, Q, T, B, fdtype, P_cfg = bench.cfg_fields(cfg)
    ih, iw, sizes = S.PYRAMIDS[pyr]
    dev = torch.device('cuda:0')
    model = bench.build_model(T, len(sizes), dev, P_cfg)
    feats = S.make_features(B, T, sizes, seed=0, device=dev, dtype=fdtype)
    if fdtype != torch.float32:
        feats = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]
    bbox, qfeat = [t.to(dev) for t in S.make_queries(B, Q, seed=0)]
    metas = S.make_img_metas(B, T, ih, iw)
    return lambda: model(bbox, qfeat, list(feats), None, metas)


def isolated(raw, M, gap_s, n=12, zero=False):
    """the two launches alone at the decoder's operand shapes for M rows: device idle for gap_s before each launch.
    zero: all-zero operands -- the SAME instruction stream with (almost) no data toggling: if the clock is set by the power the data
    path draws and not by the instruction mix, it rises (the guide's DVFS note: zero-filled inputs ran at 2.30 vs 1.90-1.95 GHz)"""
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)      # noqa: E731
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device='cuda').manual_seed(1)
    N, K = 32768, 256
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / 16
    b = torch.randn(N, device='cuda', generator=g)
    y = torch.empty(M, N, device='cuda')
    wf, wsc = dense.pack_f16s_frags(w)
    xf, xsc = dense.pack_f16s_frags(x, per_tensor=True)
    xo = torch.randn(M, N, device='cuda', generator=g).clamp_min(0)
    wo = torch.randn(K, N, device='cuda', generator=g) / N ** 0.5
    bo = torch.randn(K, device='cuda', generator=g)
    wof, wosc = dense.pack_f16s_frags(wo)
    xp = dense.f16s_pairs(xo, 9)
    if zero:                   # the packed operand images themselves (the scales stay those of the random tensors: finite)
        for t in (wf, xf, wof, xp, b, bo):
            t.zero_()
    res = {}
    for name, kinds, call in (('generator', (1, 0), lambda: lib.sbev_linear_f16s_gen(p(xf), p(xsc), p(wf), p(wsc[1]), p(b), p(y), M, N, K, N, 0, 3, st)),
                              ('out_projection', (3, 2), lambda: dense.linear_splitk_f16s(xp, wof, wosc, bo, nprod=3, x_up_log2=9, x_is_pairs=True))):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        rows = []
        for _ in range(n):
            time.sleep(gap_s)
            raw.sbev_debug_wgtime_clear()
            call()
            torch.cuda.synchronize()
            for k in kinds:
                r = read_kind(raw, k)
                if r:
                    r['kernel'] = KINDS[k]
                    rows.append(r)
                    break
        if rows:
            res[name] = {'kernel': rows[0]['kernel'], 'launches': len(rows),
                         'shader_clock_ghz_median_of_launches': round(float(np.median([r['shader_clock_ghz_median'] for r in rows])), 3),
                         'shader_clock_ghz_min_max': [min(r['shader_clock_ghz_median'] for r in rows), max(r['shader_clock_ghz_median'] for r in rows)],
                         'launch_span_us_median': round(float(np.median([r['launch_span_us'] for r in rows])), 2)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r5_gemm_clock.json'))
    ap.add_argument('--only', default=None, choices=('steady',), help='steady: only the in-step condition (for runs beneath rocprofv3 --pmc)')
    ap.add_argument('--configs', default='c2,c3')
    ap.add_argument('--label', default='un-profiled')
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    raw = raw_lib()
    out = {'label': args.label, 'lib': os.path.relpath(_lib.LIB_PATH, ROOT), 'device': torch.cuda.get_device_name(0),
           'note': 'shader clock = s_memtime cycles / s_memrealtime wall time (100 MHz) between the first and the last instruction of every '
                   'workgroup of the LAST launch of each kernel; median / p10 / p90 over its workgroups', 'steady': {}, 'isolated': {}}
    smp = None
    if args.only is None:
        smp = Sysfs()
        out['hwmon_listing'] = smp.listing
        smp.start()
        smp.mark('idle_before')
        time.sleep(1.0)
    for config in args.configs.split(','):
        if smp:
            smp.mark('setup_' + config)         # (model + feature generation, warm-up: not a measurement window)
        step = decoder_step(config)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        if smp:
            time.sleep(0.5)
            smp.mark('idle_before_' + config)
            time.sleep(0.5)
            smp.mark('busy_steady_' + config)
        raw.sbev_debug_wgtime_clear()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 1.5:         # back to back, long enough for the power management to settle
            for _ in range(20):
                step()
            n += 20
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if smp:
            smp.mark('after_' + config)
        rows = {KINDS[k]: read_kind(raw, k) for k in KINDS}
        out['steady'][config] = {'steps': n, 'ms_per_step_incl_syncs': round(1e3 * dt / n, 4), 'kernels': {k: v for k, v in rows.items() if v}}
        del step
        torch.cuda.empty_cache()
    if args.only is None:
        smp.mark('idle_between')
        time.sleep(1.0)
        smp.mark('busy_isolated')
        for config, M in (('c2', 900), ('c3', 3200)):
            if config in args.configs.split(','):
                out['isolated'][config] = {'gap_ms_before_each_launch': 5.0, **isolated(raw, M, 0.005)}
                out.setdefault('isolated_zero_operands', {})[config] = {'gap_ms_before_each_launch': 5.0, **isolated(raw, M, 0.005, zero=True)}
        smp.mark('idle_after')
        time.sleep(1.0)
        smp.mark('end')
        smp.stop_flag = True
        smp.join()
        out['sysfs'] = smp.summary()
        out['smi_snapshot_idle_after'] = smi_snapshot()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, 'w'), indent=1)
    for cond in ('steady', 'isolated', 'isolated_zero_operands'):
        for config, d in out.get(cond, {}).items():
            ks = d['kernels'] if cond == 'steady' else {v['kernel']: v for v in d.values() if isinstance(v, dict)}
            for k, v in ks.items():
                print('%-22s %-3s %-56s %s GHz  span %s us' % (cond, config, k[:56], v.get('shader_clock_ghz_median', v.get('shader_clock_ghz_median_of_launches')), v.get('launch_span_us', v.get('launch_span_us_median'))))
    if 'sysfs' in out:
        for name, w in out['sysfs']['windows'].items():
            print('sysfs %-20s %s' % (name, {k: v for k, v in w.items() if k not in ('samples', 'seconds')}))


if __name__ == '__main__':
    main()
