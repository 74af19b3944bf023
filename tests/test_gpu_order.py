"""Launch order of the fused gather + mixing items (sbev_query_order, sbev_sample_mix_*_ordered, sbev_decoder_query_order): the
order is a placement hint -- which XCD runs an item, and when -- so (1) the sort must produce a permutation of each sample's rows,
sorted by the direction of the box centre, whatever the boxes hold; (2) the fused launch must be BIT-identical under any
permutation; (3) the decoder runtime, eager and as a replayed step graph, must give identical outputs with the ordering on and
off.  The reference has no counterpart (its CUDA op runs in launch order, msmv_sampling_forward.cu:75-164)."""
import copy
import math

import pytest
import torch

from sparsebev_amd import _lib, ops, runtime as rt, synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PREFIX = 'decoder.decoder_layer.'


def centre_angle(bbox, pc_range):
    x = bbox[..., 0].double() * (pc_range[3] - pc_range[0]) + pc_range[0]
    y = bbox[..., 1].double() * (pc_range[4] - pc_range[1]) + pc_range[1]
    return torch.remainder(torch.atan2(y, x), 2 * math.pi)            # 0 = +x, counter-clockwise: the order of the kernel's key


@pytest.mark.parametrize('B,Q,ld', [(1, 900, 10), (3, 400, 10), (2, 1600, 10), (1, 1, 10), (2, 37, 4), (1, 4096, 2)])
def test_query_order_is_a_permutation_sorted_by_centre_direction(B, Q, ld):
    g = torch.Generator(device=DEV).manual_seed(Q + B)
    bbox = torch.rand(B, Q, ld, generator=g, device=DEV)
    order = ops.query_order(bbox, S.PC_RANGE).reshape(B, Q).long()
    for b in range(B):
        rows = order[b] - b * Q
        assert rows.min() >= 0 and rows.max() < Q
        assert torch.equal(torch.sort(rows).values, torch.arange(Q, device=DEV))
        ang = centre_angle(bbox[b], S.PC_RANGE)[rows]
        # monotone up to the rounding of the fp32 key (neighbouring angles closer than that may swap)
        assert (ang[1:] - ang[:-1]).min().item() > -1e-5 if Q > 1 else True
    # deterministic
    assert torch.equal(ops.query_order(bbox, S.PC_RANGE).reshape(B, Q).long(), order)


def test_query_order_head_grid_and_degenerate_boxes():
    # the head's own initialisation (BEV raster): every direction bucket of the ring is hit in order
    bbox, _ = S.make_queries(1, 900, seed=0)
    order = ops.query_order(bbox.to(DEV), S.PC_RANGE).long()
    ang = centre_angle(bbox[0], S.PC_RANGE).to(DEV)[order]
    assert (ang[1:] - ang[:-1]).min().item() > -1e-5 and ang[0] < 0.1 and ang[-1] > 2 * math.pi - 0.1
    # NaN / Inf centres, all rows identical, a centre exactly on the ego origin: still a permutation (the key is compared as an integer)
    bad = torch.rand(2, 300, 10, device=DEV)
    bad[0, ::7, 0] = float('nan')
    bad[0, 3::11, 1] = float('inf')
    bad[0, 5, 0:2] = 0.5
    bad[1] = 0.25
    o = ops.query_order(bad, S.PC_RANGE).reshape(2, 300).long()
    for b in range(2):
        assert torch.equal(torch.sort(o[b] - b * 300).values, torch.arange(300, device=DEV))
    assert torch.equal(o[1] - 300, torch.arange(300, device=DEV))          # ties keep the row order (the row index is the low key half)


def test_query_order_argument_checks():
    lib = _lib.load()
    assert lib.sbev_query_order_max() == 4096
    with pytest.raises(RuntimeError):
        ops.query_order(torch.zeros(1, 4097, 10, device=DEV), S.PC_RANGE)
    with pytest.raises(RuntimeError):
        ops.query_order(torch.zeros(1, 8, 1, device=DEV), S.PC_RANGE)
    assert ops.query_order(torch.zeros(0, 8, 10, device=DEV), S.PC_RANGE).numel() == 0


@pytest.mark.parametrize('B,Q,T,pyr,dtype,P', [(1, 900, 8, 'tiny', torch.float32, 4), (2, 37, 4, 'tiny5', torch.float32, 4),
                                               (3, 5, 16, 'tiny', torch.bfloat16, 4), (1, 64, 12, 'r50_704x256', torch.float32, 4),
                                               (1, 30, 5, 'tiny5', torch.float32, 4), (1, 50, 15, 'tiny5', torch.bfloat16, 8),
                                               (2, 21, 15, 'tiny', torch.float32, 8), (1, 1, 8, 'tiny', torch.float32, 4)])
def test_fused_launch_is_bit_identical_under_any_order(B, Q, T, pyr, dtype, P):
    ih, iw, sizes = S.PYRAMIDS[pyr]
    L, G, C = len(sizes), 4, 64
    g = torch.Generator(device=DEV).manual_seed(B * 100 + Q + T)
    levels = [torch.randn(B * T * 6, h, w, G * C, generator=g, device=DEV).to(dtype) for h, w in sizes]
    loc = torch.rand(B * T * G, Q, P, 3, generator=g, device=DEV) * 1.3 - 0.15
    loc[..., 2] = torch.randint(0, 6, (B * T * G, Q, P), generator=g, device=DEV).float() / 5
    w = torch.softmax(torch.randn(B * T * G, Q, P, L, generator=g, device=DEV), -1)
    params = torch.randn(B, Q, G * (C * C + 128 * T * P), generator=g, device=DEV) * 0.3
    want = ops.sample_mix(levels, B, T, G, loc, w, params, 128)
    bbox = torch.rand(B, Q, 10, generator=g, device=DEV)
    sorted_order = ops.query_order(bbox, S.PC_RANGE)
    shuffled = torch.randperm(B * Q, generator=g, device=DEV).int()             # not even sample-contiguous: still every row exactly once
    identity = torch.arange(B * Q, device=DEV, dtype=torch.int32)
    for order in (sorted_order, shuffled, identity):
        got = ops.sample_mix(levels, B, T, G, loc, w, params, 128, order=order)
        assert torch.equal(got, want)
    with pytest.raises(RuntimeError):
        ops.sample_mix(levels, B, T, G, loc, w, params, 128, order=identity.long())
    with pytest.raises(RuntimeError):
        ops.sample_mix(levels, B, T, G, loc, w, params, 128, order=identity[:-1] if B * Q > 1 else torch.zeros(2, device=DEV, dtype=torch.int32))


@pytest.mark.parametrize('T,L,pyr,P,B,Q,gemm', [(8, 4, 'tiny', 4, 2, 100, 'f16x3'), (4, 5, 'tiny5', 4, 1, 225, 'f32'), (15, 5, 'tiny5', 8, 1, 64, 'f16x3'),
                                                (8, 4, 'r50_704x256', 4, 1, 900, 'f16x3')])
def test_decoder_runtime_ordered_equals_launch_order(T, L, pyr, P, B, Q, gemm):
    ih, iw, sizes = S.PYRAMIDS[pyr]
    params = S.make_params(3, embed_dims=256, num_frames=T, num_points=P, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=P, num_layers=3, num_levels=L, pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    m.decoder.gemm_mode = gemm
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=4)]
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=5)]
    metas = S.make_img_metas(B, T, ih, iw)
    prev = rt.query_order(False)
    try:
        with torch.no_grad():
            a = [t.clone() for t in m(bbox, feat, list(feats), None, copy.deepcopy(metas))]
            rt.query_order(True)
            outs = [[t.clone() for t in m(bbox, feat, list(feats), None, copy.deepcopy(metas))] for _ in range(4)]     # eager, capture, replays
            r = m.decoder._runtime
            assert r.launches_per_layer(B, Q) == 7
            assert r.step_graphs.replays >= 1
            # mode 2 (untested): ONE sort per step, from the step's input boxes; every layer walks that order -- eager, captured, replayed
            rt.query_order(2)
            assert rt._STATE['order'] == 2 and r.launches_per_layer(B, Q) == 6
            outs += [[t.clone() for t in m(bbox, feat, list(feats), None, copy.deepcopy(metas))] for _ in range(4)]
            rt.query_order(False)
            assert r.launches_per_layer(B, Q) == 6
            c = [t.clone() for t in m(bbox, feat, list(feats), None, copy.deepcopy(metas))]
    finally:
        rt.query_order(prev)
    for o in outs + [c]:
        assert torch.equal(o[0], a[0]) and torch.equal(o[1], a[1])
    torch.cuda.synchronize()


def test_ordered_step_graph_with_fresh_tensors_and_a_busy_second_stream():
    """the side-stream sort inside a replayed graph: 30 replays with newly allocated inputs while another stream keeps the GPU busy;
    every replay equals the launch-order result of the same inputs"""
    B, Q, T, L = 1, 400, 8, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    params = S.make_params(5, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=6, num_levels=L, pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    feats0 = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=7)]
    metas = S.make_img_metas(B, T, ih, iw)
    side = torch.cuda.Stream()
    junk = torch.randn(2048, 2048, device=DEV)
    prev = rt.query_order(False)
    try:
        with torch.no_grad():
            want = {}
            for seed in range(3):
                bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=seed)]
                want[seed] = [t.clone() for t in m(bbox, feat, [f.clone() for f in feats0], None, copy.deepcopy(metas))]
            rt.query_order(True)
            for it in range(30):
                seed = it % 3
                bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=seed)]
                with torch.cuda.stream(side):
                    junk = junk @ junk * 1e-3
                got = m(bbox, feat, [f.clone() for f in feats0], None, copy.deepcopy(metas))
                assert torch.equal(got[0], want[seed][0]) and torch.equal(got[1], want[seed][1]), it
            assert m.decoder._runtime.step_graphs.replays >= 20
    finally:
        rt.query_order(prev)
    torch.cuda.synchronize()
