"""The fused gather + adaptive-mixing launch (sbev_sample_mix_f32) against the two launches it replaces: the workgroup of
a (query, group) item samples its own [T*P, 64] rows with the sampler's chunk code and mixes them, so results must be
BIT-identical to msmv_sampling (OUT_MIX) followed by the mixing kernel -- dense and ring pyramids, fp32 and bf16 storage,
4 and 5 levels -- and the decoder runtime must give identical outputs with the fusion on and off."""
import copy
import ctypes

import pytest
import torch

from sparsebev_amd import _lib, ops, runtime as rt, synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PREFIX = 'decoder.decoder_layer.'


def mixing(x, params, out_points=128):
    B, Q, G, Pin, C = x.shape
    y = torch.empty(B, Q, G * out_points * C, device=x.device)
    st = _lib.load().sbev_adaptive_mixing_f32(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(params.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                              B * Q, G, Pin, C, out_points, 1e-5, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    return y


@pytest.mark.parametrize('B,Q,T,pyr,dtype,P', [(1, 900, 8, 'tiny', torch.float32, 4), (2, 37, 4, 'tiny5', torch.float32, 4),
                                               (1, 100, 8, 'tiny5', torch.bfloat16, 4), (3, 5, 16, 'tiny', torch.bfloat16, 4),
                                               (1, 64, 12, 'r50_704x256', torch.float32, 4),
                                               # round 3: in_points not a multiple of 16 (12, 20, 60), 8 points per frame (two chunks),
                                               # the 15-frame x 8-point shape (120 in_points: 8 row tiles, gathered rows on the S buffer)
                                               (2, 20, 3, 'tiny', torch.float32, 4), (1, 30, 5, 'tiny5', torch.float32, 4),
                                               (1, 33, 15, 'tiny', torch.bfloat16, 4), (1, 30, 8, 'tiny', torch.float32, 8),
                                               (1, 50, 15, 'tiny5', torch.bfloat16, 8), (2, 21, 15, 'tiny', torch.float32, 8),
                                               (1, 40, 15, 'tiny5', torch.float32, 8),
                                               # round 4: S through LDS in two k halves (more than 64 in-points); 116 in-points = a second half of 13 pieces per row
                                               (1, 20, 29, 'tiny', torch.float32, 4), (2, 9, 29, 'tiny5', torch.bfloat16, 4),
                                               # round 4: fp16 feature storage
                                               (1, 100, 8, 'tiny', torch.float16, 4), (2, 37, 4, 'tiny5', torch.float16, 4),
                                               (1, 50, 15, 'tiny5', torch.float16, 8), (1, 64, 8, 'r50_704x256', torch.float16, 4)])
def test_fused_launch_is_bit_identical_to_sampler_then_mixing(B, Q, T, pyr, dtype, P):
    ih, iw, sizes = S.PYRAMIDS[pyr]
    L, G, C = len(sizes), 4, 64
    g = torch.Generator(device=DEV).manual_seed(B * 100 + Q + T)
    levels = [torch.randn(B * T * 6, h, w, G * C, generator=g, device=DEV).to(dtype) for h, w in sizes]
    loc = torch.rand(B * T * G, Q, P, 3, generator=g, device=DEV) * 1.3 - 0.15          # incl. a border band and outside points
    loc[..., 2] = torch.randint(0, 6, (B * T * G, Q, P), generator=g, device=DEV).float() / 5
    w = torch.softmax(torch.randn(B * T * G, Q, P, L, generator=g, device=DEV), -1)
    params = torch.randn(B, Q, G * (C * C + 128 * T * P), generator=g, device=DEV) * 0.3
    assert ops.sample_mix_supported(L, C, P, T, G)
    x = ops.msmv_sampling_nhwc(levels, B, T, G, loc, w, out_layout=ops.OUT_MIX)
    want = mixing(x, params)
    got = ops.sample_mix(levels, B, T, G, loc, w, params, 128)
    assert torch.equal(got, want)
    assert got.abs().max() > 0


@pytest.mark.parametrize('pyr,dtype', [('tiny', torch.float32), ('tiny5', torch.bfloat16), ('tiny', torch.float16)])
def test_fused_launch_with_nonfinite_border_pixels(pyr, dtype):
    """Inf in every border pixel of the finest level: the fused kernel's gather (buffer-load taps) must poison exactly the items the
    two launches poison, every other item is bit-identical."""
    B, Q, T, P = 1, 200, 4, 4
    ih, iw, sizes = S.PYRAMIDS[pyr]
    L, G, C = len(sizes), 4, 64
    g = torch.Generator(device=DEV).manual_seed(77)
    levels = [torch.randn(B * T * 6, h, w, G * C, generator=g, device=DEV) for h, w in sizes]
    f = levels[0]                                                                        # the finest level: its interior is not border
    f[:, 0], f[:, -1], f[:, :, 0], f[:, :, -1] = float('inf'), float('inf'), float('inf'), float('inf')
    levels = [f.to(dtype) for f in levels]
    loc = torch.rand(B * T * G, Q, P, 3, generator=g, device=DEV) * 0.4 + 0.3            # interior ...
    loc[:, ::3] = loc[:, ::3] * 4 - 1.5                                                  # ... every third query: from far outside to the border
    loc[..., 2] = torch.randint(0, 6, (B * T * G, Q, P), generator=g, device=DEV).float() / 5
    w = torch.softmax(torch.randn(B * T * G, Q, P, L, generator=g, device=DEV), -1)
    params = torch.randn(B, Q, G * (C * C + 128 * T * P), generator=g, device=DEV) * 0.3
    x = ops.msmv_sampling_nhwc(levels, B, T, G, loc, w, out_layout=ops.OUT_MIX)
    want = mixing(x, params)
    got = ops.sample_mix(levels, B, T, G, loc, w, params, 128)
    # an item whose sampled rows hold an Inf is NaN after its first LayerNorm and exactly 0 after the ReLU behind it (v_max_f32 returns
    # the non-NaN operand; torch.relu would keep the NaN -- the reference nan_to_num's the decoder output in the end): the poisoned
    # items are the all-zero ones, and they must be the same items in both paths
    dead = (want.reshape(B, Q, G, -1) == 0).all(-1)
    assert 0.01 < dead.float().mean() < 0.9, dead.float().mean()
    assert torch.equal((got.reshape(B, Q, G, -1) == 0).all(-1), dead)
    assert torch.isfinite(got).all() and torch.equal(got, want)
    # the interior queries (never near a border) are untouched by the planted pixels
    assert not dead[:, 1::3].any() and not dead[:, 2::3].any()

def test_fused_launch_on_the_frame_ring():
    B, Q, T, n_slots, G, P, C = 2, 50, 4, 6, 4, 4, 64
    ih, iw, sizes = S.PYRAMIDS['tiny']
    L = len(sizes)
    g = torch.Generator(device=DEV).manual_seed(5)
    levels = [torch.randn(B * n_slots * 6, h, w, G * C, generator=g, device=DEV) for h, w in sizes]
    slots = [4, 0, 5, 2]
    loc = torch.rand(B * T * G, Q, P, 3, generator=g, device=DEV)
    loc[..., 2] = torch.randint(0, 6, (B * T * G, Q, P), generator=g, device=DEV).float() / 5
    w = torch.softmax(torch.randn(B * T * G, Q, P, L, generator=g, device=DEV), -1)
    params = torch.randn(B, Q, G * (C * C + 128 * T * P), generator=g, device=DEV) * 0.3
    x = ops.msmv_sampling_ring(levels, B, T, G, slots, n_slots, loc, w)
    got = ops.sample_mix(levels, B, T, G, loc, w, params, 128, frame_slots=slots, n_slots=n_slots)
    assert torch.equal(got, mixing(x, params))


def test_unsupported_shapes_are_refused_and_the_runtime_falls_back():
    assert not ops.sample_mix_supported(4, 64, 6, 2, 4)        # P = 6: not whole 4-point chunks
    assert not ops.sample_mix_supported(4, 64, 8, 12, 4)       # T*P = 96: 5 .. 7 row tiles are not instantiated
    assert not ops.sample_mix_supported(3, 64, 4, 8, 4)
    assert ops.sample_mix_supported(4, 64, 8, 2, 4) and ops.sample_mix_supported(5, 64, 8, 15, 4) and ops.sample_mix_supported(4, 64, 4, 2, 4)
    levels = [torch.zeros(12, 4, 4, 256, device=DEV) for _ in range(3)]
    with pytest.raises(RuntimeError):
        ops.sample_mix(levels, 1, 2, 4, torch.zeros(8, 3, 4, 3, device=DEV), torch.zeros(8, 3, 4, 3, device=DEV),
                       torch.zeros(1, 3, 4 * (4096 + 128 * 8), device=DEV), 128)


@pytest.mark.parametrize('T,L,pyr,P', [(8, 4, 'tiny', 4), (4, 5, 'tiny5', 4), (2, 4, 'tiny', 4), (15, 5, 'tiny5', 8), (3, 4, 'tiny', 4)])
def test_decoder_runtime_fused_equals_unfused(T, L, pyr, P):
    B, Q = 2, 100
    ih, iw, sizes = S.PYRAMIDS[pyr]
    params = S.make_params(3, embed_dims=256, num_frames=T, num_points=P, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=P, num_layers=3, num_levels=L, pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=4)]
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=5)]
    metas = S.make_img_metas(B, T, ih, iw)
    try:
        rt.fuse_sample_mix(True)
        a = m(bbox, feat, list(feats), None, copy.deepcopy(metas))
        rt.fuse_sample_mix(False)
        b = m(bbox, feat, list(feats), None, copy.deepcopy(metas))
    finally:
        rt.fuse_sample_mix(True)
    lw = m(bbox, feat, list(feats), None, copy.deepcopy(metas), layerwise=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    from conftest import runtime_op_by_op
    c = runtime_op_by_op(m, bbox, feat, list(feats), None, copy.deepcopy(metas), exact_gemm=True)
    assert torch.equal(c[0], lw[0]) and torch.equal(c[1], lw[1])
