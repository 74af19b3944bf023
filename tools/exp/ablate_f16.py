import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebev_amd import _lib, dense
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e3
M = 900
N, K = 32768, 256
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / 16; b = torch.randn(N, device='cuda'); y = torch.empty(M, N, device='cuda')
wf, wsc = dense.pack_f16s_frags(w); xf, xsc = dense.pack_f16s_frags(x, per_tensor=True)
g = t(lambda: lib.sbev_linear_f16s_gen(p(xf), p(xsc), p(wf), p(wsc[1]), p(b), p(y), M, N, K, N, 0, 3, st))
N, K = 256, 32768
x = torch.randn(M, K, device='cuda').clamp_min(0); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
wf, wsc = dense.pack_f16s_frags(w); xp = dense.f16s_pairs(x, 9)
nscale = torch.empty(N, device='cuda'); lib.sbev_f16s_out_scale(p(wsc[1].contiguous()), 9, p(nscale), N, st)
plan = lib.sbev_linear_bf16s_out_plan(M, N, K); ws = torch.empty(plan * M * N, device='cuda'); y = torch.empty(M, N, device='cuda')
o1 = t(lambda: lib.sbev_linear_splitk_f16s(p(xp), 1, 9, p(wf), p(nscale), p(b), None, None, None, 1e-5, p(y), M, N, K, K, 0, 3, p(ws), st))
o0 = t(lambda: lib.sbev_linear_splitk_f16s(p(x), 0, 9, p(wf), p(nscale), p(b), None, None, None, 1e-5, p(y), M, N, K, K, 0, 3, p(ws), st))
print('%-28s gen %6.1f us   out(pairs) %6.1f us   out(split in kernel) %6.1f us  (out incl. reducer)' % (os.path.basename(_lib.LIB_PATH), g, o1, o0))
