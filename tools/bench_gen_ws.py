"""Dev tool: the weight-stationary generator kernel against the tiled ping-pong kernel (csrc/gemm_bf16s.hip), f16x3, K = 256:
HIP-event median per launch at the decoder's shapes, bit identity, error vs fp64 at c2.  python tools/bench_gen_ws.py [--m 900 ...]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebev_amd import _lib, dense   # noqa: E402

lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def t(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


ap = argparse.ArgumentParser()
ap.add_argument('--shapes', nargs='*', default=['900x32768', '3200x32768', '3600x32768', '1600x77824', '7200x32768'])
args = ap.parse_args()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
K = 256
for sh in args.shapes:
    M, N = [int(v) for v in sh.split('x')]
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    wf, wsc = dense.pack_f16s_frags(w)
    xs, xsc = dense.pack_f16s_frags(x, per_tensor=True)
    y = torch.empty(M, N, device='cuda')
    fn = lambda: lib.sbev_linear_f16s_gen(p(xs), p(xsc), p(wf), p(wsc[1]), p(b), p(y), M, N, K, N, 0, 3, st)
    out = {}
    for name, on in (('ws', 1), ('tiled', 0)):
        lib.sbev_linear_gen_weight_stationary(on)
        fn(); torch.cuda.synchronize()
        out[name] = (y.clone(), t(fn))
    lib.sbev_linear_gen_weight_stationary(1)
    same = torch.equal(out['ws'][0], out['tiled'][0])
    err = ''
    if M <= 1000:
        ref = x.double() @ w.double().t() + b.double()
        err = ' max err vs fp64 %.2e' % (out['ws'][0].double() - ref).abs().max().item()
    fl = 3 * 2.0 * M * N * K
    print('gen %5d x %5d: ws %6.1f us (min %6.1f, %4.0f TF fp16)  tiled %6.1f us (min %6.1f)  bit-identical %s%s'
          % (M, N, out['ws'][1][0], out['ws'][1][1], fl / out['ws'][1][0] / 1e6, out['tiled'][1][0], out['tiled'][1][1], same, err))
