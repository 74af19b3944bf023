"""GPU parity of the sampling path, called through the C ABI (sparsebev_amd.ops -> libsbev_hip.so):
HIP kernels vs golden vectors made by the reference, vs the CPU oracle on seeded inputs, and -- at the
full BASELINE config-2 size -- vs the C oracle plus size-independent properties.
Tolerance: 1e-4 abs fp32 on sampled features (north_star); the camera-hit mask, selected view and
projected coordinates must be bit-identical."""
import numpy as np
import pytest
import torch

from conftest import load_golden, feats_of
from sparsebev_amd import _lib, ops, synthetic as S

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = 'cuda:0'


def dev(t):
    return t.to(DEV)


@pytest.mark.parametrize('tag', ['L4_C8', 'L4_C64', 'L5_C8', 'L5_C64', 'L4_C16_P7'])
def test_g1_msmv_vs_reference_golden(tag):
    g = load_golden('g1_msmv_' + tag)
    feats = [dev(f) for f in feats_of(g)]
    out = ops.msmv_sampling(feats, dev(g['loc']), dev(g['weights']))
    assert out.shape == g['out'].shape
    assert (out.cpu() - g['out']).abs().max() < TOL
    # mixing-ready layout is the same numbers, permuted (T=G=1 -> [B',Q,1,P,C])
    mix = ops.msmv_sampling(feats, dev(g['loc']), dev(g['weights']), out_layout=ops.OUT_MIX, T=1, G=1)
    assert torch.equal(mix[:, :, 0].permute(0, 1, 3, 2), out)


def test_msmv_bf16_features_fp32_accumulate():
    from oracle import sparsebev_oracle as O
    g = load_golden('g1_msmv_L4_C64')
    feats_bf = [f.to(torch.bfloat16) for f in feats_of(g)]
    ref = O.msmv_sampling_kernel_semantics([f.float() for f in feats_bf], g['loc'], g['weights'])
    out = ops.msmv_sampling([dev(f) for f in feats_bf], dev(g['loc']), dev(g['weights']))
    assert (out.cpu() - ref).abs().max() < TOL       # bf16 is storage only: exact widening, fp32 math


@pytest.mark.parametrize('tag', ['L4_C64', 'L5_C64', 'L4_C16_P7'])
def test_msmv_fp16_features_fp32_accumulate(tag):
    """fp16 STORAGE (round 4; what the reference's fp16 eval mode has before its out_fp32 cast, models/sparsebev.py:46): the taps are
    widened exactly, so the result equals the fp32 kernel's on the widened features BIT for bit, both tap paths, both output layouts"""
    from oracle import sparsebev_oracle as O
    g = load_golden('g1_msmv_' + tag)
    feats_h = [f.to(torch.float16) for f in feats_of(g)]
    ref = O.msmv_sampling_kernel_semantics([f.float() for f in feats_h], g['loc'], g['weights'])
    prev = _lib.load().sbev_msmv_buffer_taps(1)
    try:
        for buf in (1, 0):
            _lib.load().sbev_msmv_buffer_taps(buf)
            out = ops.msmv_sampling([dev(f) for f in feats_h], dev(g['loc']), dev(g['weights']))
            assert (out.cpu() - ref).abs().max() < TOL
            assert torch.equal(out, ops.msmv_sampling([dev(f.float()) for f in feats_h], dev(g['loc']), dev(g['weights'])))
            mix = ops.msmv_sampling([dev(f) for f in feats_h], dev(g['loc']), dev(g['weights']), out_layout=ops.OUT_MIX, T=1, G=1)
            assert torch.equal(mix[:, :, 0].permute(0, 1, 3, 2), out)
    finally:
        _lib.load().sbev_msmv_buffer_taps(prev)


@pytest.mark.parametrize('T', [1, 8])
def test_g2_projection_bit_exact_and_sampling4d(T):
    from oracle import sparsebev_oracle as O
    g = load_golden('g2_sampling4d_T%d' % T)
    pts, sw = g['sample_points'], g['scale_weights']
    B, Q, _, G, P, _ = pts.shape
    ih, iw = [int(v) for v in g['image_hw']]
    loc, uvh, valid, iview = ops.project_select(dev(pts.reshape(B, Q, T, G * P, 3)), dev(g['lidar2img']), ih, iw, G, P, dump=True)
    assert torch.equal(valid.cpu(), g['valid'])                                                    # camera-hit mask
    assert np.array_equal(uvh.cpu().numpy().view(np.uint32), g['uvh'].numpy().view(np.uint32))      # DUMP tap, bitwise
    assert torch.equal(iview.cpu().long(), torch.argmax(g['valid'].permute(0, 1, 3, 4, 2), dim=-1))
    _, taps = O.sampling_4d(pts, O.regroup_features(feats_of(g), False), sw, g['lidar2img'], ih, iw, O.msmv_sampling_gridsample)
    assert np.array_equal(loc.cpu().numpy().view(np.uint32), taps['loc_bp'].contiguous().numpy().view(np.uint32))
    # full sampling_4d through the HIP sampler (reference weight order incl. quirk q1 prepared on host here)
    feats_cl = [dev(f) for f in O.regroup_features(feats_of(g), True)]
    out = ops.msmv_sampling(feats_cl, loc, dev(taps['w_bp'].contiguous()), out_layout=ops.OUT_MIX, T=T, G=G)
    assert (out.cpu() - g['out']).abs().max() < TOL


def test_g3_front_project_gather_chain():
    g = load_golden('g3_sampling_T2')
    B, Q, T, L = [int(v) for v in g['cfg']]
    G, P = 4, 4
    seeds = [int(v) for v in g['seeds']]
    params = S.make_params(seeds[0], embed_dims=256, num_frames=T, num_points=P, num_levels=L)
    feats = S.make_features(B, T, [tuple(int(x) for x in s) for s in g['sizes']], seed=seeds[2])
    ih, iw = [int(v) for v in g['image_hw']]
    qf = g['query_feat']
    off = torch.nn.functional.linear(qf, params['sampling.sampling_offset.weight'], params['sampling.sampling_offset.bias'])
    lg = torch.nn.functional.linear(qf, params['sampling.scale_weights.weight'], params['sampling.scale_weights.bias'])
    pts, wbp = ops.sampling_front(dev(g['query_bbox']), dev(off), dev(lg), dev(g['time_diff']), S.PC_RANGE, T, G, P, L)
    loc = ops.project_select(pts, dev(g['lidar2img']), ih, iw, G, P)
    # zero-copy NHWC pyramid: [B*T*6, H, W, 256]
    nhwc = [dev(f.reshape(B * T * 6, 256, *f.shape[-2:]).permute(0, 2, 3, 1).contiguous()) for f in feats]
    out = ops.msmv_sampling_nhwc(nhwc, B, T, G, loc, wbp)
    assert out.shape == g['out'].shape
    assert (out.cpu() - g['out']).abs().max() < TOL


def test_front_kernel_vs_oracle():
    from oracle import sparsebev_oracle as O
    B, Q, T, G, P, L = 2, 100, 8, 4, 4, 5
    bbox, feat = S.make_queries(B, Q, seed=5)
    params = S.make_params(5, num_frames=T, num_levels=L)
    td = torch.tensor([[0.0, 0.5, 1.0, 1.5, 2.1, 2.5, 3.0, 3.7], [0.0, 0.4, 1.0, 1.5, 2.0, 2.5, 3.2, 3.5]])
    pts_ref, sw_ref = O.sampling_front(params, bbox, feat, td, S.PC_RANGE, T, P, L)
    off = torch.nn.functional.linear(feat, params['sampling.sampling_offset.weight'], params['sampling.sampling_offset.bias'])
    lg = torch.nn.functional.linear(feat, params['sampling.scale_weights.weight'], params['sampling.scale_weights.bias'])
    pts, wbp = ops.sampling_front(dev(bbox), dev(off), dev(lg), dev(td), S.PC_RANGE, T, G, P, L)
    assert (pts.cpu() - pts_ref.reshape(B, Q, T, G * P, 3)).abs().max() < 1e-4      # metres
    w_ref = sw_ref.reshape(B, Q, G, T, P, L).permute(0, 2, 3, 1, 4, 5).reshape(B * G * T, Q, P, L)
    assert (wbp.cpu() - w_ref).abs().max() < 1e-6


def test_msmv_edge_cases():
    f = [torch.randn(2, 6, 4, 5, 8, device=DEV)]
    # empty query set and empty batch
    assert ops.msmv_sampling(f, torch.zeros(2, 0, 4, 3, device=DEV), torch.zeros(2, 0, 4, 1, device=DEV)).shape == (2, 0, 8, 4)
    # P at the reference's MAX_POINT
    loc = torch.rand(2, 3, 32, 3, device=DEV)
    loc[..., 2] = torch.randint(0, 6, (2, 3, 32), device=DEV).float() / 5
    w = torch.rand(2, 3, 32, 1, device=DEV)
    from oracle import sparsebev_oracle as O
    ref = O.msmv_sampling_kernel_semantics([f[0].cpu()], loc.cpu(), w.cpu())
    assert (ops.msmv_sampling(f, loc, w).cpu() - ref).abs().max() < TOL
    with pytest.raises(RuntimeError, match='num_point exceed limits'):
        ops.msmv_sampling(f, torch.rand(2, 3, 33, 3, device=DEV), torch.rand(2, 3, 33, 1, device=DEV))
    with pytest.raises(RuntimeError, match='contiguous'):
        ops.msmv_sampling([f[0].transpose(2, 3)], loc, w)
    # NaN / inf / far-away coordinates: contribute exactly zero, no fault
    bad = torch.tensor([float('nan'), float('inf'), -float('inf'), 1e30, -1e30, 0.5], device=DEV)
    loc = torch.stack([bad, bad.flip(0), torch.full_like(bad, 0.2)], -1).reshape(1, 1, 6, 3).repeat(2, 1, 1, 1).contiguous()
    out = ops.msmv_sampling(f, loc, torch.ones(2, 1, 6, 1, device=DEV))
    assert torch.isfinite(out).all() and out.abs().sum() == 0


def c2_inputs(B=1, Q=900, T=8, seed=0):
    ih, iw, sizes = S.PYRAMIDS['r50_704x256']
    G, P, L = 4, 4, len(sizes)
    g = torch.Generator(device=DEV).manual_seed(seed)
    feats = [torch.randn(B * T * G, 6, h, w, 64, generator=g, device=DEV) for h, w in sizes]
    bbox, feat = S.make_queries(B, Q, seed=seed)
    params = S.make_params(seed, num_frames=T, num_levels=L)
    metas = S.make_img_metas(B, T, ih, iw)
    l2i = torch.from_numpy(np.asarray([m['lidar2img'] for m in metas]).astype(np.float32))
    td = torch.tensor([[0.5 * t for t in range(T)]] * B)
    off = torch.nn.functional.linear(feat, params['sampling.sampling_offset.weight'], params['sampling.sampling_offset.bias'])
    lg = torch.nn.functional.linear(feat, params['sampling.scale_weights.weight'], params['sampling.scale_weights.bias'])
    pts, wbp = ops.sampling_front(dev(bbox), dev(off), dev(lg), dev(td), S.PC_RANGE, T, G, P, L)
    loc, uvh, valid, iview = ops.project_select(pts, dev(l2i), ih, iw, G, P, dump=True)
    return feats, pts, l2i, loc, wbp, (uvh, valid, iview), (ih, iw, B, Q, T, G, P, L)


def test_full_size_c2_vs_c_oracle_and_properties():
    """BASELINE config 2 (r50 704x256, 900 q, T=8, bs=1): 115 200 sampled points per layer."""
    from oracle import c_oracle
    feats, pts, l2i, loc, wbp, (uvh, valid, iview), (ih, iw, B, Q, T, G, P, L) = c2_inputs()
    # (1) projection + mask bit-exact against the C oracle at full size
    uvh_r, valid_r, iview_r = c_oracle.project(pts.cpu().numpy(), l2i.numpy(), ih, iw)
    assert np.array_equal(valid.cpu().numpy(), valid_r)
    assert np.array_equal(iview.cpu().numpy(), iview_r)
    assert np.array_equal(uvh.cpu().numpy().view(np.uint32), uvh_r.view(np.uint32))
    hit = valid_r.sum(2)
    assert 0.85 < (hit >= 1).mean() < 0.99 and (hit >= 2).mean() > 0.01      # the rig exercises 0/1/2-hit points
    # (2) sampler vs the C oracle at full size
    out = ops.msmv_sampling(feats, loc, wbp)
    ref = c_oracle.msmv_fwd([f.cpu().numpy() for f in feats], loc.cpu().numpy(), wbp.cpu().numpy())
    assert np.abs(out.cpu().numpy() - ref).max() < TOL
    # (3) size-independent properties
    #  linearity in the features
    feats2 = [torch.randn_like(f) for f in feats]
    o2 = ops.msmv_sampling(feats2, loc, wbp)
    o12 = ops.msmv_sampling([1.5 * a - 0.5 * b for a, b in zip(feats, feats2)], loc, wbp)
    assert (o12 - (1.5 * out - 0.5 * o2)).abs().max() < 5e-5
    #  constant features + weights summing to one -> every fully-inside point returns the constant
    ones = [torch.ones_like(f) for f in feats]
    oc = ops.msmv_sampling(ones, loc, wbp)
    inside = ((loc[..., 0] > 0) & (loc[..., 0] < 1) & (loc[..., 1] > 0) & (loc[..., 1] < 1))   # [B',Q,P]
    assert (oc.permute(0, 1, 3, 2)[inside] - 1).abs().max() < 1e-5
    #  layouts agree bit for bit; determinism
    mix = ops.msmv_sampling(feats, loc, wbp, out_layout=ops.OUT_MIX, T=T, G=G)
    ref_mix = out.reshape(B, T, G, Q, 64, P).permute(0, 3, 2, 1, 5, 4).reshape(B, Q, G, T * P, 64)
    assert torch.equal(mix, ref_mix)
    assert torch.equal(ops.msmv_sampling(feats, loc, wbp), out)


@pytest.mark.parametrize('tag', ['L4_C64', 'L5_C64'])
@pytest.mark.parametrize('buffer_taps', [1, 0])
def test_g12_nonfinite_border_pixels_match_the_cuda_kernel_semantics(tag, buffer_taps):
    """Inf / NaN planted in border pixels (fixture G12).  The reference never reads an out-of-map bilinear corner
    (msmv_sampling_forward.cu:47-66): a point wholly outside a map gets exactly 0 from it, a point whose footprint covers a bad pixel
    turns non-finite.  The HIP sampler -- buffer-load taps (out-of-map = out-of-range offset, hardware zeros) and the 64-bit
    global-load path (select) -- must give the non-finite elements of the kernel-semantics oracle EXACTLY (positions and kind),
    the finite ones to 1e-4, in both output layouts and for bf16 storage; the reference's own output pins the positions."""
    from conftest import assert_same_with_nonfinite
    from oracle import sparsebev_oracle as O
    from sparsebev_amd import _lib
    g = load_golden('g12_msmv_nonfinite_' + tag)
    feats_cl = feats_of(g)
    want = O.msmv_sampling_kernel_semantics(feats_cl, g['loc'], g['weights'])
    prev = _lib.load().sbev_msmv_buffer_taps(buffer_taps)
    try:
        feats = [dev(f) for f in feats_cl]
        out = ops.msmv_sampling(feats, dev(g['loc']), dev(g['weights']))
        assert_same_with_nonfinite(out, want, TOL, 'HIP vs kernel semantics')
        assert_same_with_nonfinite(out, g['out'], TOL, 'HIP vs the reference', kinds=False)
        mix = ops.msmv_sampling(feats, dev(g['loc']), dev(g['weights']), out_layout=ops.OUT_MIX, T=1, G=1)
        a, b = mix[:, :, 0].permute(0, 1, 3, 2), out
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))
        feats_bf = [f.to(torch.bfloat16) for f in feats_cl]
        want_bf = O.msmv_sampling_kernel_semantics([f.float() for f in feats_bf], g['loc'], g['weights'])
        out_bf = ops.msmv_sampling([dev(f) for f in feats_bf], dev(g['loc']), dev(g['weights']))
        assert_same_with_nonfinite(out_bf, want_bf, TOL, 'HIP bf16 storage vs kernel semantics')
    finally:
        _lib.load().sbev_msmv_buffer_taps(prev)


def test_points_outside_every_map_ignore_nonfinite_pixels_at_full_size():
    """Config-2 shape with every border pixel of every map set to Inf: the sample points of the synthetic rig that hit no camera
    (7.7 %, arbitrary coordinates) and any other point wholly outside a level must come out exactly as with zeroed borders."""
    feats, pts, l2i, loc, wbp, _, (ih, iw, B, Q, T, G, P, L) = c2_inputs()
    bad, zero = [f.clone() for f in feats], [f.clone() for f in feats]
    for fb, fz in zip(bad, zero):
        for t, v in ((fb, float('inf')), (fz, 0.0)):
            t[:, :, 0], t[:, :, -1], t[:, :, :, 0], t[:, :, :, -1] = v, v, v, v
    o_bad, o_zero = ops.msmv_sampling(bad, loc, wbp), ops.msmv_sampling(zero, loc, wbp)
    fin = torch.isfinite(o_bad)
    assert 0.5 < fin.float().mean() < 1.0                       # most points are interior; some really cover a border pixel
    assert torch.equal(o_bad[fin], o_zero[fin])
    # a point is finite exactly when none of its in-map corners is a border pixel: check the "wholly outside" ones explicitly
    x, y = loc[..., 0], loc[..., 1]                             # [B', Q, P]
    H0, W0 = feats[0].shape[2:4]
    far = (x < -1.0 / (8 - 1)) | (x > 1 + 1.0 / (8 - 1)) | (y < -1.0 / (8 - 1)) | (y > 1 + 1.0 / (8 - 1))   # outside even the coarsest map's one-pixel band
    assert far.any()
    assert (o_bad.permute(0, 1, 3, 2)[far] == 0).all()


def test_int64_offsets_beyond_2g_elements():
    """The reference's int32 offsets overflow once B'*N*H*W*C >= 2^31.  Sample the
    LAST batch entry of a 2.2e9-element level and check it against a small tensor holding just that entry."""
    H, W, C, N = 128, 352, 64, 6
    Bp = 128                                     # 128*6*128*352*64 = 2.21e9 elements = 8.9 GB fp32
    try:
        big = torch.empty(Bp, N, H, W, C, device=DEV)
    except RuntimeError:
        pytest.skip('not enough device memory for the 8.9 GB level')
    g = torch.Generator(device=DEV).manual_seed(1)
    last = torch.randn(1, N, H, W, C, generator=g, device=DEV)
    big[-1] = last[0]
    Q, P = 64, 4
    loc = torch.rand(1, Q, P, 3, generator=g, device=DEV)
    loc[..., 2] = torch.randint(0, N, (1, Q, P), generator=g, device=DEV).float() / (N - 1)
    w = torch.rand(1, Q, P, 1, generator=g, device=DEV)
    small = ops.msmv_sampling([last], loc, w)
    loc_big = torch.zeros(Bp, Q, P, 3, device=DEV)
    w_big = torch.zeros(Bp, Q, P, 1, device=DEV)
    loc_big[-1], w_big[-1] = loc[0], w[0]
    out = ops.msmv_sampling([big], loc_big, w_big)
    assert torch.equal(out[-1], small[0])


@pytest.mark.parametrize('tag', ['L4_C8', 'L5_C64'])
@torch.enable_grad()
def test_g8_msmv_backward_vs_reference_autograd(tag):
    """grads of the HIP op vs the reference's native-PyTorch sampler differentiated by autograd."""
    g = load_golden('g8_msmv_bwd_' + tag)
    feats = [dev(f).requires_grad_(True) for f in feats_of(g)]
    loc, w = dev(g['loc']).requires_grad_(True), dev(g['weights']).requires_grad_(True)
    out = ops.msmv_sampling(feats, loc, w)
    out.backward(dev(g['grad_out']))
    for i, f in enumerate(feats):
        assert (f.grad.cpu() - g['grad_feat%d' % i]).abs().max() < TOL
    assert (w.grad.cpu() - g['grad_weights']).abs().max() < TOL
    scale = max(1.0, g['grad_loc_xy'].abs().max().item())
    assert (loc.grad.cpu()[..., :2] - g['grad_loc_xy']).abs().max() < TOL * scale
    assert loc.grad[..., 2].abs().max() == 0            # like the reference op: no gradient for the view index


@torch.enable_grad()
def test_msmv_backward_full_size_vs_oracle_sample():
    """Config-2-sized backward (115 200 points): spot-check against the torch oracle on one sample batch entry."""
    from oracle import sparsebev_oracle as O
    feats, pts, l2i, loc, wbp, _, (ih, iw, B, Q, T, G, P, L) = c2_inputs()
    gout = torch.randn(loc.shape[0], Q, 64, P, device=DEV)
    fl = [f.clone().requires_grad_(True) for f in feats]
    lc, ww = loc.clone().requires_grad_(True), wbp.clone().requires_grad_(True)
    ops.msmv_sampling(fl, lc, ww).backward(gout)
    b = 5
    gf, gl, gw = O.msmv_sampling_backward([f[b:b + 1].cpu() for f in feats], loc[b:b + 1].cpu(), wbp[b:b + 1].cpu(), gout[b:b + 1].cpu())
    # 1e-4 of each gradient's own scale (the north-star tolerance, relative: these gradients are sums of up to hundreds of
    # O(1) terms -- grad_loc multiplies by the map size, a level-3 pixel collects every tap of its 22 x 8 map -- so an absolute
    # 1e-4 would be below fp32 resolution of the values themselves; summation order differs: in-wave tree / float atomics
    # against the oracle's index_add)
    def rel(a, r):
        return ((a - r).abs().max() / r.abs().max().clamp_min(1.0)).item()
    assert rel(ww.grad[b:b + 1].cpu(), gw) < 1e-4, rel(ww.grad[b:b + 1].cpu(), gw)
    assert rel(lc.grad[b:b + 1].cpu(), gl) < 1e-4, rel(lc.grad[b:b + 1].cpu(), gl)
    for a, r in zip(fl, gf):
        assert rel(a.grad[b:b + 1].cpu(), r) < 1e-4, (rel(a.grad[b:b + 1].cpu(), r), r.abs().max().item())


@pytest.mark.parametrize('P,L,C', [(6, 4, 64), (1, 5, 64), (9, 2, 64), (5, 3, 24)])
@torch.enable_grad()
def test_msmv_backward_point_tails_and_both_kernels_vs_oracle(P, L, C):
    """Backward with point counts that are not multiples of the 4-point chunk (C = 64 fast kernel) and a channel count
    that takes the generic kernel; coordinates include out-of-map taps and exact grid points."""
    from oracle import sparsebev_oracle as O
    g = torch.Generator().manual_seed(P * 10 + L)
    sizes = [(9, 14), (5, 7), (3, 4), (2, 2), (1, 3)][:L]
    Bp, Q = 3, 10
    feats = [torch.randn(Bp, 6, h, w, C, generator=g) for h, w in sizes]
    loc = torch.rand(Bp, Q, P, 3, generator=g) * 1.3 - 0.15
    loc[..., 2] = torch.randint(0, 6, (Bp, Q, P), generator=g).float() / 5
    loc[0, 0, 0, :2] = torch.tensor([0.0, 1.0])
    loc[0, 1, 0, :2] = torch.tensor([0.5, 0.5])
    wts = torch.softmax(torch.randn(Bp, Q, P, L, generator=g), -1)
    gout = torch.randn(Bp, Q, C, P, generator=g)
    fl = [f.to(DEV).requires_grad_(True) for f in feats]
    lc, ww = loc.to(DEV).requires_grad_(True), wts.to(DEV).requires_grad_(True)
    ops.msmv_sampling(fl, lc, ww).backward(gout.to(DEV))
    gf, gl, gw = O.msmv_sampling_backward(feats, loc, wts, gout)
    assert (ww.grad.cpu() - gw).abs().max() < 1e-4
    assert (lc.grad.cpu() - gl).abs().max() < 1e-4 * max(1.0, gl.abs().max().item())
    for a_, r in zip(fl, gf):
        assert (a_.grad.cpu() - r).abs().max() < 1e-4


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_pipelined_items_equal_single_items(dtype):
    """Above 8192 (b', q) items the kernel walks two items per wave with the second one's coordinates prefetched; below
    it handles one per wave.  Same inputs through both code paths (one big call vs the same queries in slices small
    enough to take the one-item path) must agree bit for bit, also for an odd item count and bf16 storage."""
    g = torch.Generator().manual_seed(99)
    sizes = [(12, 20), (6, 10), (3, 5), (2, 3)]
    Bp, Q, P, L, C = 7, 1999, 4, 4, 64                      # 13 993 items: odd, > 8192
    feats = [torch.randn(Bp, 6, h, w, C, generator=g).to(DEV).to(dtype) for h, w in sizes]
    loc = torch.rand(Bp, Q, P, 3, generator=g) * 1.2 - 0.1
    loc[..., 2] = torch.randint(0, 6, (Bp, Q, P), generator=g).float() / 5
    wts = torch.softmax(torch.randn(Bp, Q, P, L, generator=g), -1)
    loc, wts = loc.to(DEV), wts.to(DEV)
    big = ops.msmv_sampling(feats, loc, wts)
    parts = [ops.msmv_sampling(feats, loc[:, s:s + 500].contiguous(), wts[:, s:s + 500].contiguous()) for s in range(0, Q, 500)]
    assert all(p.shape[0] * p.shape[1] < 8192 for p in parts)
    assert torch.equal(big, torch.cat(parts, dim=1))


def test_g10_old_box_convention_front_kernel_and_decoder_layer():
    """VERSION.name = 'v0.17.1' (old checkpoints, val.py:128-129): the sample-point kernel against the reference recording
    (G10) and one decoder layer + get_bboxes against the oracle under the same switch."""
    import copy
    from oracle import sparsebev_oracle as O
    from sparsebev_amd.utils import VERSION
    from sparsebev_amd.transformer import SparseBEVTransformer
    from sparsebev_amd import head as H
    g = load_golden('g10_sample_points_versions')
    B, Q, GP = g['offset'].shape[:3]
    td = torch.zeros(B, 1)
    logits = torch.zeros(B, Q, GP * 4)
    try:
        for name, key in (('v1.0.0', 'pts_v1'), ('v0.17.1', 'pts_v017')):
            VERSION.name = name
            pts, _ = ops.sampling_front(g['query_bbox'].to(DEV), g['offset'].reshape(B, Q, GP * 3).to(DEV), logits.to(DEV), td.to(DEV),
                                        S.PC_RANGE, T=1, G=4, P=4, L=4)
            assert (pts.cpu().reshape(B, Q, GP, 3) - g[key]).abs().max() < 1e-4, name
        VERSION.name = O.VERSION_NAME = 'v0.17.1'
        T, L, Qn = 2, 4, 36
        ih, iw, sizes = S.PYRAMIDS['tiny']
        params = S.make_params(81, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
        m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=1, num_levels=L, num_classes=10, code_size=10, pc_range=S.PC_RANGE)
        m.load_state_dict({'decoder.decoder_layer.' + k: v for k, v in params.items()})
        m = m.to(DEV).eval()
        bbox, feat = S.make_queries(1, Qn, seed=82)
        feats = S.make_features(1, T, sizes, seed=83)
        metas = S.make_img_metas(1, T, ih, iw)
        cls, box = m(bbox.to(DEV), feat.to(DEV), [f.to(DEV) for f in feats], None, copy.deepcopy(metas))
        rc, rb, _ = O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=1)
        assert (cls[0].cpu() - rc[0]).abs().max() < TOL and (box[0].cpu() - rb[0]).abs().max() < TOL
        O.VERSION_NAME = 'v1.0.0'
        rc1, _, _ = O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=1)
        assert (rc1[0] - rc[0]).abs().max() > 1e-3                       # the switch matters
        O.VERSION_NAME = 'v0.17.1'
        # get_bboxes: w / l swapped, yaw -> -yaw - pi/2
        post = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]
        gen = torch.Generator().manual_seed(84)
        c2 = torch.randn(1, 1, 50, 10, generator=gen)
        b2 = torch.randn(1, 1, 50, 10, generator=gen)
        coder = H.NMSFreeCoder(S.PC_RANGE, post_center_range=post, max_num=20, score_threshold=None, num_classes=10)
        got = coder._decode({'all_cls_scores': c2.to(DEV), 'all_bbox_preds': b2.to(DEV)}, True)[0]
        ref = O.get_bboxes(O.nms_free_decode(c2, b2, 10, 20, None, post))[0]
        assert torch.equal(got['labels'].cpu(), ref[2]) and (got['bboxes'].cpu() - ref[0]).abs().max() < 2e-5
    finally:
        VERSION.name = O.VERSION_NAME = 'v1.0.0'
