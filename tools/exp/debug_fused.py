import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sparsebev_amd import _lib, ops, synthetic as S
DEV='cuda:0'
def mixing(x, params, out_points=128):
    B, Q, G, Pin, C = x.shape
    y = torch.empty(B, Q, G * out_points * C, device=x.device)
    st = _lib.load().sbev_adaptive_mixing_f32(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(params.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                              B * Q, G, Pin, C, out_points, 1e-5, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    return y
with torch.no_grad():
  for (B,Q,T,pyr,dtype) in [(1, 900, 8, 'tiny', torch.float32), (2, 37, 4, 'tiny5', torch.float32), (1, 37, 4, 'tiny5', torch.float32), (2, 37, 4, 'tiny', torch.float32), (2, 37, 8, 'tiny5', torch.float32),
                          (1, 100, 8, 'tiny5', torch.bfloat16), (3, 5, 16, 'tiny', torch.bfloat16), (1, 64, 12, 'r50_704x256', torch.float32)]:
    ih, iw, sizes = S.PYRAMIDS[pyr]
    L, G, P, C = len(sizes), 4, 4, 64
    g = torch.Generator(device=DEV).manual_seed(B * 100 + Q + T)
    levels = [torch.randn(B * T * 6, h, w, G * C, generator=g, device=DEV).to(dtype) for h, w in sizes]
    loc = torch.rand(B * T * G, Q, P, 3, generator=g, device=DEV) * 1.3 - 0.15
    loc[..., 2] = torch.randint(0, 6, (B * T * G, Q, P), generator=g, device=DEV).float() / 5
    w = torch.softmax(torch.randn(B * T * G, Q, P, L, generator=g, device=DEV), -1)
    params = torch.randn(B, Q, G * (C * C + 128 * T * P), generator=g, device=DEV) * 0.3
    x = ops.msmv_sampling_nhwc(levels, B, T, G, loc, w, out_layout=ops.OUT_MIX)
    want = mixing(x, params)
    got = ops.sample_mix(levels, B, T, G, loc, w, params, 128)
    d = (got - want).abs().view(B, Q, G, -1).amax(-1)
    print((B,Q,T,pyr,dtype), 'equal', torch.equal(got, want), 'max diff %.3e' % d.max().item(), 'items differing', int((d > 0).sum()), 'of', d.numel(), 'first', (d > 0).nonzero()[:3].tolist())
