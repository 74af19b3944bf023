"""Per-phase cycle stamps of the ping-pong out-projection (variant library built with -DSBEV_EXP_TRACE)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebev_amd import _lib, dense
M, N, K, nimg = 900, 256, 32768, 3
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
x = torch.randn(M, K, device='cuda').clamp_min(0); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
if mode == 'f16x3':
    wf, wsc = dense.pack_f16s_frags(w)
    xp = dense.f16s_pairs(x, 9) if len(sys.argv) > 2 else None
    for _ in range(3):
        if xp is not None:
            dense.linear_splitk_f16s(xp, wf, wsc, b, nprod=3, x_up_log2=9, x_is_pairs=True)
        else:
            dense.linear_splitk_f16s(x, wf, wsc, b, nprod=3, x_up_log2=9)
else:
    wp = dense.pack_bf16s_frags(w, nimg)
    for _ in range(3):
        dense.linear_splitk_bf16s(x, wp, b, nimg=nimg)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 512 * 8))()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.sbev_debug_trace_read.argtypes = [ctypes.c_void_p]
assert raw.sbev_debug_trace_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 512, 8).astype(np.int64)
for grp in (0, 1):
    print('half', grp, ' (cycles) frag reads | W loads | split+stage+X loads | barrier | MFMAs (+W loads) | barrier | -> next')
    for g in list(range(0, 6)) + [14, 15, 28, 29]:
        r = t[grp, g]
        nxt = t[grp, g + 1, 0]
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - max(r[3], r[1]), r[5] - r[4], r[6] - r[5], nxt - r[6]]
        print('  s=%2d' % g, ' '.join('%6d' % v for v in d), '  total', nxt - r[0])
    print('  first stamp .. last stamp of 30 slabs:', t[grp, 29, 6] - t[grp, 0, 0], 'cycles')
wt = (ctypes.c_ulonglong * (1024 * 4))()
raw.sbev_debug_wgtime_read.argtypes = [ctypes.c_void_p, ctypes.c_int]      # (round 5: one table per kernel kind, tools/gemm_clock.py)
if raw.sbev_debug_wgtime_read(wt, 3 if mode == 'f16x3' else 2) == 0:
    a = np.array(wt, dtype=np.uint64).reshape(1024, 4).astype(np.int64)[:256]
    a = a[a[:, 2] > 0]
    if a[:, 2].max() > 0:
        us = (a[:, 2] - a[:, 0]) / 100.0
        cyc = a[:, 3] - a[:, 1]
        print('workgroup lifetime us: median %.1f min %.1f max %.1f; shader clocks %.0f -> %.2f GHz; launch span %.1f us; start skew %.1f us' % (np.median(us), us.min(), us.max(), np.median(cyc), np.median(cyc / us) / 1e3, (a[:, 2].max() - a[:, 0].min()) / 100.0, (a[:, 0].max() - a[:, 0].min()) / 100.0))
        first = t[0, 0, 0]
        print('workgroup 0: start -> first FETCH stamp %d cycles; last stamp -> end %d cycles' % (first - a[0, 1], a[0, 3] - t[0, 29, 6]))
