"""Dev tool: time the parameter-generator GEMM [M,256]x[N,256]^T over M (prologue vs per-row cost) and the
out-projection over M."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebev_amd import dense   # noqa: E402


def t(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e3


shapes = [(32768, 256, [16, 32, 64, 256, 512, 896, 900, 1024, 1800, 3600])]
if '--all' in sys.argv:
    shapes.append((256, 32768, [768, 896, 900, 1024, 1800]))
for (N, K, Ms) in shapes:
    for M in Ms:
        x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
        us = t(lambda: dense.linear(x, w, b))
        print('M=%5d N=%6d K=%6d  %8.1f us %6.1f TF  (%.3f us per row)' % (M, N, K, us, 2.0 * M * N * K / us / 1e6, us / M))
