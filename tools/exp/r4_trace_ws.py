"""Dev tool (variant library built with -DSBEV_EXP_TRACE): workgroup lifetimes and shader clock of the weight-stationary generator,
and the per-fragment cycle stamps of waves 0 (group A) and 4 (group B) of workgroup 0."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebev_amd import _lib, dense
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 900
N, K = 32768, 256
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / 16; b = torch.randn(N, device='cuda')
y = torch.empty(M, N, device='cuda')
wf, wsc = dense.pack_f16s_frags(w); xf, xsc = dense.pack_f16s_frags(x, per_tensor=True)
for _ in range(20):
    lib.sbev_linear_f16s_gen(p(xf), p(xsc), p(wf), p(wsc[1]), p(b), p(y), M, N, K, N, 0, 3, st)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (2 * 512 * 8))()
raw.sbev_debug_trace_read.argtypes = [ctypes.c_void_p]
assert raw.sbev_debug_trace_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 512, 8).astype(np.int64)
for grp in (0, 1):
    G = int((t[grp, :, 0] > 0).sum())
    print('group %s: fragment start -> [MFMAs to sync point] [vmcnt wait] [barrier] [DMA issue] -> next fragment start; cycles' % 'AB'[grp])
    for g in range(min(G, 16)):
        r = t[grp, g]
        nxt = t[grp, g + 1, 0] if g + 1 < G else 0
        print('  i=%2d  to-sync %6d  wait %5d  barrier %5d  issue %5d  rest %6d   period %6d' % (g, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], (nxt - r[4]) if nxt else 0, (nxt - r[0]) if nxt else 0))
buf2 = (ctypes.c_ulonglong * (1024 * 4))()
raw.sbev_debug_wgtime_read.argtypes = [ctypes.c_void_p, ctypes.c_int]      # (round 5: one table per kernel kind, tools/gemm_clock.py)
assert raw.sbev_debug_wgtime_read(buf2, 1) == 0
wg = np.array(buf2, dtype=np.uint64).reshape(1024, 4).astype(np.int64)
wg = wg[wg[:, 0] > 0]
t0 = wg[:, 0].min()
us = (wg[:, 2] - wg[:, 0]) / 100.0
cyc = wg[:, 3] - wg[:, 1]
print('%d workgroups: start spread %.1f us, kernel span %.1f us; lifetime us median %.1f min %.1f max %.1f; shader clock %.2f GHz (median of cycles / lifetime)'
      % (len(wg), (wg[:, 0] - t0).max() / 100.0, (wg[:, 2].max() - t0) / 100.0, np.median(us), us.min(), us.max(), np.median(cyc / us) / 1e3))
