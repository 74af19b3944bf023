"""Dev tool: the split-bf16 GEMMs (csrc/gemm_bf16s.hip) at the decoder's shapes against the exact f32-MFMA kernels and the
round-2 bf16x3 kernels: HIP-event median per launch, equivalent TFLOP/s, max error against fp64."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebev_amd import _lib, dense   # noqa: E402

lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def t(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e3


import argparse
ap = argparse.ArgumentParser()
ap.add_argument('--m', type=int, nargs='*', default=[900, 3200, 3600])
ap.add_argument('--quick', action='store_true', help='new kernels only, no fp64 reference')
args = ap.parse_args()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
if args.quick:
    for M in args.m:
        line = 'M=%4d ' % M
        for nimg in (3, 2):
            N, K = 32768, 256
            x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
            y = torch.empty(M, N, device='cuda')
            ws = dense.pack_bf16s_frags(w, nimg); xs = dense.pack_bf16s_frags(x, nimg)
            us = t(lambda: lib.sbev_linear_bf16s_gen(p(xs), p(ws), p(b), p(y), M, N, K, N, 0, nimg, st))
            line += ' gen x%d %6.1f us' % (6 if nimg == 3 else 3, us)
            N, K = 256, 32768
            x = torch.randn(M, K, device='cuda').clamp_min(0); w = torch.randn(N, K, device='cuda') / K ** 0.5
            wp = dense.pack_bf16s_frags(w, nimg)
            plan = lib.sbev_linear_bf16s_out_plan(M, N, K)
            wsb = torch.empty(plan * M * N, device='cuda'); y = torch.empty(M, N, device='cuda')
            us = t(lambda: lib.sbev_linear_splitk_bf16s(p(x), p(wp), p(b), None, None, None, 1e-5, p(y), M, N, K, K, 0, nimg, p(wsb), st))
            line += ' out x%d %6.1f us |' % (6 if nimg == 3 else 3, us)
        print(line)
    sys.exit(0)
for M in args.m:
    # generator
    N, K = 32768, 256
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    ref = x.double() @ w.double().t() + b.double()
    y = torch.empty(M, N, device='cuda')
    us = t(lambda: dense.linear(x, w, b))
    err = (dense.linear(x, w, b).double() - ref).abs().max().item()
    print('gen  M=%4d f32-mfma   %7.1f us  %6.1f TF        err %.2e' % (M, us, 2.0 * M * N * K / us / 1e6, err))
    for nimg in (3, 2):
        ws = dense.pack_bf16s_frags(w, nimg)
        xs = dense.pack_bf16s_frags(x, nimg)
        us_split = t(lambda: dense.pack_bf16s_frags(x, nimg))
        fn = lambda: lib.sbev_linear_bf16s_gen(p(xs), p(ws), p(b), p(y), M, N, K, N, 0, nimg, st)
        us = t(fn)
        err = (y.double() - ref).abs().max().item()
        npr = 6 if nimg == 3 else 3
        print('gen  M=%4d bf16x%d     %7.1f us  %6.1f TF(eq) %6.0f TF(bf16)  err %.2e   (+ split of X %.1f us incl. alloc)'
              % (M, npr, us, 2.0 * M * N * K / us / 1e6, npr * 2.0 * M * N * K / us / 1e6, err, us_split))
    del ref
    # out-projection
    N, K = 256, 32768
    x = torch.randn(M, K, device='cuda').clamp_min(0); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    ref = x.double() @ w.double().t() + b.double()
    us = t(lambda: dense.linear(x, w, b))
    err = (dense.linear(x, w, b).double() - ref).abs().max().item()
    print('out  M=%4d f32-mfma   %7.1f us  %6.1f TF        err %.2e   (incl. reducer)' % (M, us, 2.0 * M * N * K / us / 1e6, err))
    for nimg in (3, 2):
        wp = dense.pack_bf16s_frags(w, nimg)
        plan = lib.sbev_linear_bf16s_out_plan(M, N, K)
        wsb = torch.empty(plan * M * N, device='cuda')
        y = torch.empty(M, N, device='cuda')
        fn = lambda: lib.sbev_linear_splitk_bf16s(p(x), p(wp), p(b), None, None, None, 1e-5, p(y), M, N, K, K, 0, nimg, p(wsb), st)
        us = t(fn)
        err = (y.double() - ref).abs().max().item()
        npr = 6 if nimg == 3 else 3
        print('out  M=%4d bf16x%d     %7.1f us  %6.1f TF(eq) %6.0f TF(bf16)  err %.2e   (incl. reducer, %d slabs)'
              % (M, npr, us, 2.0 * M * N * K / us / 1e6, npr * 2.0 * M * N * K / us / 1e6, err, plan))
    del ref
