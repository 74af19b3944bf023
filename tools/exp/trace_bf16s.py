"""Per-phase cycle stamps of the ping-pong generator (variant library built with -DSBEV_EXP_TRACE)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebev_amd import _lib, dense
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K, nimg = 900, 32768, 256, 3
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / 16; b = torch.randn(N, device='cuda')
y = torch.empty(M, N, device='cuda')
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
if mode == 'f16x3':
    wf, wsc = dense.pack_f16s_frags(w); xf, xsc = dense.pack_f16s_frags(x, per_tensor=True)
    for _ in range(3):
        lib.sbev_linear_f16s_gen(p(xf), p(xsc), p(wf), p(wsc[1]), p(b), p(y), M, N, K, N, 0, 3, st)
else:
    ws = dense.pack_bf16s_frags(w, nimg); xs = dense.pack_bf16s_frags(x, nimg)
    for _ in range(3):
        lib.sbev_linear_bf16s_gen(p(xs), p(ws), p(b), p(y), M, N, K, N, 0, nimg, st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 512 * 8))()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.sbev_debug_trace_read.argtypes = [ctypes.c_void_p]
assert raw.sbev_debug_trace_read(buf) == 0
import numpy as np
t = np.array(buf, dtype=np.uint64).reshape(2, 512, 8).astype(np.int64)
names = ['issue', 'wait', 'reads+st', 'bar1', 'mfma', 'bar2', 'next']
for grp in (0, 1):
    print('group', grp, ' (cycles) issue | vmcnt wait | frag reads (+stores) | barrier | MFMAs | barrier | -> next FETCH')
    G = int((t[grp, :, 0] > 0).sum())
    for g in list(range(0, 6)) + list(range(14, 20)) + [G - 2]:
        r = t[grp, g]
        nxt = t[grp, g + 1, 0] if g + 1 < G else r[6]
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], nxt - r[6]]
        print('  g=%2d' % g, ' '.join('%6d' % v for v in d), '  total', nxt - r[0])
    tot = t[grp, G - 1, 6] - t[grp, 0, 0]
    print('  %d stages:' % G, tot, 'cycles ->', tot / G, 'per stage')

buf2 = (ctypes.c_ulonglong * (1024 * 4))()
raw.sbev_debug_wgtime_read.argtypes = [ctypes.c_void_p, ctypes.c_int]      # (round 5: one table per kernel kind, tools/gemm_clock.py)
assert raw.sbev_debug_wgtime_read(buf2, 0) == 0
w = np.array(buf2, dtype=np.uint64).reshape(1024, 4).astype(np.int64)
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
print('%d workgroups; realtime ticks (100 MHz = 10 ns): start min/max %d / %d, end min/max %d / %d' % (len(w), 0, (w[:, 0] - t0).max(), (w[:, 2] - t0).min(), (w[:, 2] - t0).max()))
life_rt = (w[:, 2] - w[:, 0]); life_ck = (w[:, 3] - w[:, 1])
print('lifetime realtime ticks: min %d median %d max %d ; shader-counter ticks: min %d median %d max %d ; counter ticks per realtime tick: %.1f (x 100 MHz = counter rate)'
      % (life_rt.min(), np.median(life_rt), life_rt.max(), life_ck.min(), np.median(life_ck), life_ck.max(), np.median(life_ck / np.maximum(life_rt, 1))))
order = np.argsort(w[:, 2])
print('slowest workgroups (index, lifetime rt):', [(int(i), int(life_rt[i])) for i in order[-6:]], ' fastest:', [(int(i), int(life_rt[i])) for i in order[:6]])
us = (w[:, 2] - w[:, 0]) / 100.0
cyc = w[:, 3] - w[:, 1]
print('workgroup lifetime us: median %.1f min %.1f max %.1f; %.2f GHz' % (np.median(us), us.min(), us.max(), np.median(cyc / us) / 1e3))
