"""Ablations of gemm_tn_f16s_kernel (timing only): which part of the K step the launch time is made of.
   tools/build_variant.sh exp_tn_<x> gemm_tn_f16s.hip -DSBEV_TN_NO_<X>; SBEV_LIB_PATH=... python tools/exp/ablate_tn.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebev_amd import _lib, dense
lib = _lib.load()
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e3
K = 900
res = []
for M, N in ((256, 32768), (32768, 256)):
    A, B = torch.randn(K, M, device='cuda'), torch.randn(K, N, device='cuda')
    sa, sb = dense.f16s_tensor_scale(A), dense.f16s_tensor_scale(B)
    out = torch.empty(M, N, device='cuda')
    res.append(t(lambda: dense.gemm_tn_f16s(A, M, sa, B, N, sb, M, N, K, out=out, ldc=N)))
print('%-30s  [256 x 32768] %6.1f us   [32768 x 256] %6.1f us' % (os.path.basename(_lib.LIB_PATH), res[0], res[1]))
