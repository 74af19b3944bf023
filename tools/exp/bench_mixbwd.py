"""mixing backward at config 2 (900 x 4 items, in_points 32): plain entry vs the one that also writes the per-wave maxima"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebev_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e3
BQ, G, Pin, C, Pout = 900, 4, 32, 64, 128
NP = C * C + Pout * Pin
x = torch.randn(BQ, G, Pin, C, device='cuda'); params = torch.randn(BQ, G * NP, device='cuda') * 0.2; gy = torch.randn(BQ, G * Pout * C, device='cuda')
gx, gp = torch.empty_like(x), torch.empty_like(params); imax = torch.empty(BQ * G * 4, device='cuda')
a = t(lambda: lib.sbev_adaptive_mixing_bwd_f32(p(x), p(params), p(gy), p(gx), p(gp), BQ, G, Pin, C, Pout, 1e-5, None))
b = t(lambda: lib.sbev_adaptive_mixing_bwd_max_f32(p(x), p(params), p(gy), p(gx), p(gp), p(imax), BQ, G, Pin, C, Pout, 1e-5, None))
print('%-26s mixing backward %6.1f us   with maxima %6.1f us' % (os.path.basename(_lib.LIB_PATH), a, b))
