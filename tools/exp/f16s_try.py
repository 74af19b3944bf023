"""Experiment: the fp16 hi + lo modes of gemm_bf16s.hip against bf16x6 and the exact f32-MFMA kernels: error vs fp64 and time."""
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
def errs(y, ref):
    d = (y.double() - ref).abs()
    return d.max().item(), d.pow(2).mean().sqrt().item()
def rnd(shape, seed, scale=1.0, wide=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g) * scale
    if wide: x = x * torch.exp2(torch.randint(-6, 7, shape, generator=g).float())
    return x.cuda()
for M in (900, 3200):
  for wide in (False, True):
    N, K = 32768, 256
    x, w, b = rnd((M, K), 3, wide=wide), rnd((N, K), 4, K ** -0.5, wide=wide), rnd((N,), 5)
    ref = x.double() @ w.double().t() + b.double()
    print('generator M=%d wide=%s' % (M, wide))
    print('   f32-mfma  max %.3e rms %.3e  %7.1f us' % (*errs(dense.linear(x, w, b), ref), t(lambda: dense.linear(x, w, b))))
    ws = dense.pack_bf16s_frags(w, 3); xs = dense.pack_bf16s_frags(x, 3); y = torch.empty(M, N, device='cuda')
    us = t(lambda: lib.sbev_linear_bf16s_gen(p(xs), p(ws), p(b), p(y), M, N, K, N, 0, 3, st))
    print('   bf16x6    max %.3e rms %.3e  %7.1f us' % (*errs(y, ref), us))
    wf, wsc = dense.pack_f16s_frags(w); xf, xsc = dense.pack_f16s_frags(x, per_tensor=True)
    for nprod in (3, 4):
        y.fill_(float('nan'))
        us = t(lambda: lib.sbev_linear_f16s_gen(p(xf), p(xsc), p(wf), p(wsc[1]), p(b), p(y), M, N, K, N, 0, nprod, st))
        print('   f16x%d     max %.3e rms %.3e  %7.1f us   (X scale+pack %.1f us)' % (nprod, *errs(y, ref), us, t(lambda: dense.pack_f16s_frags(x, per_tensor=True))))
    del ref
    N, K = 256, 32768
    x = rnd((M, K), 6, wide=wide).clamp_min(0); w, b = rnd((N, K), 7, K ** -0.5, wide=wide), rnd((N,), 8)
    ref = x.double() @ w.double().t() + b.double()
    print('out-proj M=%d wide=%s' % (M, wide))
    print('   f32-mfma  max %.3e rms %.3e  %7.1f us' % (*errs(dense.linear(x, w, b), ref), t(lambda: dense.linear(x, w, b))))
    wp = dense.pack_bf16s_frags(w, 3)
    print('   bf16x6    max %.3e rms %.3e  %7.1f us' % (*errs(dense.linear_splitk_bf16s(x, wp, b, nimg=3), ref), t(lambda: dense.linear_splitk_bf16s(x, wp, b, nimg=3))))
    wf, wsc = dense.pack_f16s_frags(w)
    import math
    up = 15 - math.frexp(float(x.abs().max()))[1]
    for nprod in (3, 4):
        print('   f16x%d     max %.3e rms %.3e  %7.1f us (incl. reducer + scale launch; x_up_log2 %d)' % (nprod, *errs(dense.linear_splitk_f16s(x, wf, wsc, b, nprod=nprod, x_up_log2=up), ref),
              t(lambda: dense.linear_splitk_f16s(x, wf, wsc, b, nprod=nprod, x_up_log2=up)), up))
    del ref
