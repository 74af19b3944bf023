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

# opt-in 3 x bf16 split
import ctypes
from sparsebev_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
for (M, N, K, sk) in [(900, 32768, 256, 0), (900, 256, 32768, 32), (3600, 32768, 256, 0)]:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    w2 = torch.empty(N, 2 * K, device='cuda', dtype=torch.int16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.sbev_split_bf16x3_weights(p(w), p(w2), N, K, st)
    y = torch.empty(M, N, device='cuda'); ws = torch.empty(max(sk, 1), M, N, device='cuda')
    if sk:
        fn = lambda: lib.sbev_linear_splitk_bf16x3(p(x), p(w2), p(b), None, None, None, 1e-5, p(y), M, N, K, K, 0, sk, p(ws), st)
    else:
        fn = lambda: lib.sbev_linear_bf16x3(p(x), p(w2), p(b), None, p(y), M, N, K, K, N, 0, st)
    us = t(fn)
    ref = x.double() @ w.double().t() + b.double()
    err = (y.double() - ref).abs().max().item()
    print('bf16x3 M=%5d N=%6d K=%6d  %8.1f us %6.1f TF(eq)  max err %.2e' % (M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
