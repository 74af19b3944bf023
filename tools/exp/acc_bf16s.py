"""Experiment: what sets the bf16x6 out-projection's error -- number of accumulator roundings (K chunks) or the dropped terms."""
import os, sys, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    from sparsebev_amd import dense
    M, N, K = 900, 256, 32768
    g = torch.Generator().manual_seed(6)
    x = torch.randn(M, K, generator=g).clamp_min(0).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    ref = x.double() @ w.double().t() + b.double()
    for nimg in (3, 2):
        y = dense.linear_splitk_bf16s(x, dense.pack_bf16s_frags(w, nimg), b, nimg=nimg)
        d = (y.double() - ref).abs()
        print('chunks %s nimg %d: max %.3e rms %.3e' % (sys.argv[1], nimg, d.max().item(), d.pow(2).mean().sqrt().item()))
    if sys.argv[1] == '17':
        y = dense.linear(x, w, b); d = (y.double() - ref).abs()
        print('f32-mfma: max %.3e rms %.3e' % (d.max().item(), d.pow(2).mean().sqrt().item()))
else:
    for s in (4, 8, 17, 34, 68):
        subprocess.run([sys.executable, __file__, str(s)], env=dict(os.environ, SBEV_BF16S_OUT_CHUNKS=str(s)))
