"""Dev tool: time the fp32 MFMA linear at the decoder's shapes (HIP events on the launch stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebev_amd import dense   # noqa: E402


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


for (M, N, K) in [(900, 32768, 256), (900, 256, 32768), (900, 256, 256), (900, 768, 256), (900, 512, 256), (900, 256, 512), (3600, 32768, 256), (3600, 256, 32768)]:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
    us = t(lambda: dense.linear(x, w, b))
    us_t = t(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * N * K
    print('M=%5d N=%6d K=%6d  sbev %8.1f us %6.1f TF   | rocBLAS/aten %8.1f us %6.1f TF' % (M, N, K, us, fl / us / 1e6, us_t, fl / us_t / 1e6))
