"""Pin the CPU oracle (oracle/sparsebev_oracle.py) against golden vectors produced by the REFERENCE's own
code (tests/golden/make_golden.py).  Tolerances: 1e-4 abs fp32 for sampled / mixed features (north_star),
exact for the camera-hit mask, the selected view and the projected coordinates."""
import numpy as np
import pytest
import torch

from conftest import load_golden, free_running_bound, report_free_running, feats_of
from oracle import sparsebev_oracle as O
from sparsebev_amd import synthetic as S

TOL = 1e-4


def cl_to_cf(feats_cl):
    return [f.permute(0, 4, 1, 2, 3).contiguous() for f in feats_cl]


@pytest.mark.parametrize('tag', ['L4_C8', 'L4_C64', 'L5_C8', 'L5_C64', 'L4_C16_P7'])
def test_g1_sampler_both_restatements(tag):
    g = load_golden('g1_msmv_' + tag)
    feats_cl = feats_of(g)
    a2 = O.msmv_sampling_gridsample(cl_to_cf(feats_cl), g['loc'], g['weights'])
    a1 = O.msmv_sampling_kernel_semantics(feats_cl, g['loc'], g['weights'])
    assert a2.shape == g['out'].shape
    assert (a2 - g['out']).abs().max() < 1e-6          # same algorithm as the reference: ~exact
    assert (a1 - g['out']).abs().max() < TOL           # CUDA-kernel semantics vs grid_sample: < 1e-4


@pytest.mark.parametrize('tag', ['L4_C64', 'L5_C64'])
def test_g12_nonfinite_border_pixels_all_three_restatements(tag):
    """Inf / NaN planted in border pixels, sample points from far outside to well inside (fixture G12 = the reference's own
    native-PyTorch sampler): an out-of-map bilinear corner is never read (msmv_sampling_forward.cu:47-66), so a point wholly outside
    a map gets exactly 0 from it.  The reference's output pins WHICH elements are non-finite (and the finite values); the kind
    (NaN / Inf) is the CUDA kernel's -- one view, no zero-weighted neighbour view -- on which the kernel-semantics restatement and
    the C oracle must agree exactly."""
    from conftest import assert_same_with_nonfinite
    from oracle import c_oracle
    g = load_golden('g12_msmv_nonfinite_' + tag)
    feats_cl = feats_of(g)
    assert_same_with_nonfinite(O.msmv_sampling_gridsample(cl_to_cf(feats_cl), g['loc'], g['weights']), g['out'], 1e-6, 'grid_sample')
    a1 = O.msmv_sampling_kernel_semantics(feats_cl, g['loc'], g['weights'])
    assert_same_with_nonfinite(a1, g['out'], TOL, 'kernel semantics vs the reference', kinds=False)
    ref_c = c_oracle.msmv_fwd([f.numpy() for f in feats_cl], g['loc'].numpy(), g['weights'].numpy())
    assert_same_with_nonfinite(ref_c, a1, TOL, 'C oracle vs kernel semantics')


@pytest.mark.parametrize('T', [1, 8])
def test_g2_projection_mask_bit_exact_and_quirks(T):
    g = load_golden('g2_sampling4d_T%d' % T)
    pts, sw = g['sample_points'], g['scale_weights']
    B, Q, _, G, P, _ = pts.shape
    ih, iw = [int(v) for v in g['image_hw']]
    uvh, valid = O.project_points(pts.reshape(B, Q, T, G * P, 3), g['lidar2img'], ih, iw)
    # bit-exact: compare raw fp32 bit patterns, not values
    assert torch.equal(valid.to(torch.uint8), g['valid'])
    assert np.array_equal(uvh.numpy().view(np.uint32), g['uvh'].numpy().view(np.uint32))
    nh = valid.sum(2)
    assert (nh == 0).any() and (nh == 1).any() and (nh >= 2).any()        # fixture covers 0/1/2-hit points
    feats = O.regroup_features(feats_of(g), channel_last=False)
    out, taps = O.sampling_4d(pts, feats, sw, g['lidar2img'], ih, iw, O.msmv_sampling_gridsample)
    assert (out - g['out']).abs().max() < 1e-6
    feats_cl = O.regroup_features(feats_of(g), channel_last=True)
    out_k, _ = O.sampling_4d(pts, feats_cl, sw, g['lidar2img'], ih, iw, O.msmv_sampling_kernel_semantics)
    assert (out_k - g['out']).abs().max() < TOL
    if T > 1:
        # quirk q1 is real: applying the *intended* (b,t,g) weight order gives a different answer
        L = sw.shape[-1]
        w_intended = sw.permute(0, 3, 2, 1, 4, 5).reshape(B * T * G, Q, P, L).contiguous()
        wrong = O.msmv_sampling_gridsample(feats, taps['loc_bp'].contiguous(), w_intended)
        wrong = wrong.reshape(B, T, G, Q, -1, P).permute(0, 3, 2, 1, 5, 4).reshape(out.shape)
        assert (wrong - g['out']).abs().max() > 1e-2


def _common(g):
    B, Q, T, L = [int(v) for v in g['cfg']]
    seeds = [int(v) for v in g['seeds']]
    params = S.make_params(seeds[0], embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    assert abs(S.checksum(params) - float(g['params_checksum'])) < 1e-6 * float(g['params_checksum'])
    sizes = [tuple(int(x) for x in s) for s in g['sizes']]
    feats = S.make_features(B, T, sizes, seed=seeds[2])
    assert abs(S.checksum(feats) - float(g['feats_checksum'])) < 1e-6 * float(g['feats_checksum'])
    return B, Q, T, L, params, feats


def test_g3_sampling_front_and_gather():
    g = load_golden('g3_sampling_T2')
    B, Q, T, L, params, feats = _common(g)
    ih, iw = [int(v) for v in g['image_hw']]
    pts, sw = O.sampling_front(params, g['query_bbox'], g['query_feat'], g['time_diff'], S.PC_RANGE, T, 4, L)
    out, _ = O.sampling_4d(pts, O.regroup_features(feats, False), sw, g['lidar2img'], ih, iw, O.msmv_sampling_gridsample)
    assert (out - g['out']).abs().max() < TOL


@pytest.mark.parametrize('name', ['g4_mixing_T2', 'g4_mixing_T8'])
def test_g4_adaptive_mixing(name):
    g = load_golden(name)
    T = 2 if name.endswith('T2') else 8
    params = S.make_params(int(g['seeds'][0]), embed_dims=256, num_frames=T, num_points=4, num_levels=4)
    assert abs(S.checksum(params) - float(g['params_checksum'])) < 1e-6 * float(g['params_checksum'])
    out = O.adaptive_mixing(params, g['x'], g['query_feat'])
    assert (out - g['out']).abs().max() < TOL


def test_g5_self_attention():
    g = load_golden('g5_selfattn_T2')
    B, Q, T, L, params, _ = _common(g)
    a = O.self_attention(params, g['query_bbox'], g['query_feat'], S.PC_RANGE, None)
    b = O.self_attention(params, g['query_bbox'], g['query_feat'], S.PC_RANGE, g['mask'].bool())
    assert (a - g['out_nomask']).abs().max() < TOL
    assert (b - g['out_mask']).abs().max() < TOL
    assert (g['out_mask'] - g['out_nomask']).abs().max() > 1e-3


def test_g6_decoder_layer():
    g = load_golden('g6_layer_T2')
    B, Q, T, L, params, feats = _common(g)
    ih, iw = [int(v) for v in g['image_hw']]
    x, cls, box = O.decoder_layer(params, g['query_bbox'], g['query_feat'], O.regroup_features(feats, False),
                                  g['time_diff'], g['lidar2img'], ih, iw, S.PC_RANGE, T, 4, L,
                                  O.msmv_sampling_gridsample)
    assert (x - g['out_feat']).abs().max() < TOL
    assert (cls - g['out_cls']).abs().max() < TOL
    assert (box - g['out_bbox']).abs().max() < TOL


@pytest.mark.parametrize('tag', ['c1', 'c2small', 'L5'])
def test_g7_full_decoder(tag):
    g = load_golden('g7_decoder_' + tag)
    B, Q, T, L = [int(v) for v in g['cfg']]
    seeds = [int(v) for v in g['seeds']]
    ih, iw, sizes = S.PYRAMIDS[str(g['pyramid'])]
    params = S.make_params(seeds[0], embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    feats = S.make_features(B, T, sizes, seed=seeds[2])
    assert abs(S.checksum(feats) - float(g['feats_checksum'])) < 1e-6 * float(g['feats_checksum'])
    metas = S.make_img_metas(B, T, ih, iw)
    for b, m in enumerate(metas):
        m['img_timestamp'] = [float(v) for v in g['timestamps'][b]]
    forced = forced_layer_inputs(g)
    for sampler in (O.msmv_sampling_gridsample, O.msmv_sampling_kernel_semantics):
        # teacher-forced: every layer starts from the reference's own (bbox, feat) -> single-layer budget
        cls, box, feat = O.decoder(params, g['query_bbox'], g['query_feat'], feats, metas, S.PC_RANGE,
                                   sampler=sampler, forced_inputs=forced)
        assert (cls - g['out_cls']).abs().max() < TOL
        assert (box - g['out_bbox']).abs().max() < TOL
        assert (feat - g['out_feat']).abs().max() < TOL
    # free-running 6 layers: a random-init decoder on white-noise features amplifies fp32 rounding noise layer by layer (5-8x at
    # c1's real pyramid).  The bound is the reference's OWN drift on these inputs (fixture G13: its two samplers against each other,
    # and a one-ulp nudge of query_feat), x 2 -- for both of the oracle's samplers; the kernel-semantics run is additionally held
    # against the reference's own MSMV_CUDA-path recording in G13
    bound = free_running_bound(tag)
    y = load_golden('g13_yardstick_' + tag)
    for sampler in (O.msmv_sampling_gridsample, O.msmv_sampling_kernel_semantics):
        cls, box, feat = O.decoder(params, g['query_bbox'], g['query_feat'], feats, metas, S.PC_RANGE, sampler=sampler)
        assert (cls[0] - g['out_cls'][0]).abs().max() < TOL
        report_free_running('oracle/' + sampler.__name__[14:], tag, (cls, box, feat), (g['out_cls'], g['out_bbox'], g['out_feat']), bound)
        if sampler is O.msmv_sampling_kernel_semantics:
            report_free_running('oracle-vs-ref-kernel-path', tag, (cls, box, feat), (y['kernel_cls'], y['kernel_bbox'], y['kernel_feat']), bound)


def forced_layer_inputs(g):
    """(query_bbox, query_feat) the reference fed to each of its 6 layers, from the G7 fixture."""
    n = g['out_cls'].shape[0]
    return [(g['query_bbox'], g['query_feat'])] + [(g['out_bbox'][i - 1], g['out_feat'][i - 1]) for i in range(1, n)]


# ---- the C half of the oracle (oracle/msmv_oracle.c) against the same golden vectors ----------------------
@pytest.mark.parametrize('tag', ['L4_C8', 'L5_C64', 'L4_C16_P7'])
def test_c_oracle_sampler(tag):
    from oracle import c_oracle
    g = load_golden('g1_msmv_' + tag)
    out = c_oracle.msmv_fwd([f.numpy() for f in feats_of(g)], g['loc'].numpy(), g['weights'].numpy())
    assert np.abs(out - g['out'].numpy()).max() < TOL


@pytest.mark.parametrize('T', [1, 8])
def test_c_oracle_projection_bit_exact(T):
    from oracle import c_oracle
    g = load_golden('g2_sampling4d_T%d' % T)
    pts = g['sample_points']
    B, Q, _, G, P, _ = pts.shape
    ih, iw = [int(v) for v in g['image_hw']]
    uvh, valid, iview = c_oracle.project(pts.reshape(B, Q, T, G * P, 3).numpy(), g['lidar2img'].numpy(), ih, iw)
    assert np.array_equal(valid, g['valid'].numpy())
    assert np.array_equal(uvh.view(np.uint32), g['uvh'].numpy().view(np.uint32))
    assert np.array_equal(iview, np.argmax(g['valid'].numpy(), axis=2))


@pytest.mark.parametrize('tag', ['L4_C8', 'L5_C64'])
def test_g8_sampler_backward_restatement(tag):
    g = load_golden('g8_msmv_bwd_' + tag)
    feats = feats_of(g)
    gf, gl, gw = O.msmv_sampling_backward(feats, g['loc'], g['weights'], g['grad_out'])
    for i, f in enumerate(gf):
        assert (f - g['grad_feat%d' % i]).abs().max() < TOL
    assert (gw - g['grad_weights']).abs().max() < TOL
    scale = max(1.0, g['grad_loc_xy'].abs().max().item())
    assert (gl[..., :2] - g['grad_loc_xy']).abs().max() < TOL * scale
    assert gl[..., 2].abs().max() == 0


@pytest.mark.parametrize('tag', ['c2', 'small', 'few'])
def test_g9_nms_free_decode_restatement(tag):
    """oracle.nms_free_decode / denormalize_bbox vs the reference's NMSFreeCoder.decode run on the same tensors."""
    g = load_golden('g9_nms_free_' + tag)
    B, Q, NC, max_num = [int(v) for v in g['cfg']]
    thr = float(g['thr'])
    thr = None if thr < 0 else thr
    assert torch.equal(O.denormalize_bbox(g['box'][-1]), g['denorm_all'])
    dec = O.nms_free_decode(g['cls'], g['box'], NC, max_num, thr, [float(v) for v in g['post']])
    assert len(dec) == B
    for i, d in enumerate(dec):
        assert torch.equal(d['labels'], g['labels%d' % i])
        assert torch.equal(d['scores'], g['scores%d' % i])
        assert torch.equal(d['bboxes'], g['bboxes%d' % i])
    kept = sum(len(d['scores']) for d in dec)
    assert 0 < kept < B * max_num            # both masks are exercised


def test_head_prepare_and_postprocess_restatement():
    """Shapes / values of the head's eval-branch query init and of its output re-formatting (no golden: the head class
    itself needs mmdet's DETRHead; the six arithmetic lines are restated from models/sparsebev_head.py:85-95)."""
    g = torch.Generator().manual_seed(5)
    init, lab = torch.rand(16, 10, generator=g), torch.randn(11, 255, generator=g)
    qb, qf = O.head_prepare(init, lab, 10, 3)
    assert qb.shape == (3, 16, 10) and qf.shape == (3, 16, 256)
    assert torch.equal(qb[2], init) and torch.equal(qf[1, 7, :255], lab[10]) and float(qf[..., 255].abs().max()) == 0.0
    box = torch.rand(2, 3, 16, 10, generator=g)
    out = O.head_postprocess(box, S.PC_RANGE)
    assert torch.equal(out[..., 0], box[..., 0] * 102.4 + (-51.2)) and torch.equal(out[..., 4], box[..., 2] * 8.0 + (-5.0))
    assert torch.equal(out[..., 2:4], box[..., 3:5]) and torch.equal(out[..., 5:], box[..., 5:])


def test_g10_sample_points_both_box_conventions():
    """make_sample_points under VERSION 'v1.0.0' and 'v0.17.1' (rotation sign, models/utils.py:66-77) vs the reference."""
    g = load_golden('g10_sample_points_versions')
    try:
        for name, key in (('v1.0.0', 'pts_v1'), ('v0.17.1', 'pts_v017')):
            O.VERSION_NAME = name
            got = O.make_sample_points(g['query_bbox'], g['offset'], S.PC_RANGE)
            assert (got - g[key]).abs().max() < 1e-5, name
    finally:
        O.VERSION_NAME = 'v1.0.0'


@pytest.mark.parametrize('tag,tol', [('L2', 2e-4), ('L6', 5e-2)])
def test_g11_oracle_autograd_matches_the_reference_gradients(tag, tol):
    """The oracle is built from differentiable torch ops; its autograd gradients must reproduce fixture G11 = the reference
    decoder's own train()-mode gradients (mmcv dropouts at 0).  This pins the oracle as the gradient checker the GPU
    backward tests use (tests/test_gpu_backward.py)."""
    from sparsebev_amd import synthetic as S
    g = load_golden('g11_train_' + tag)
    B, Q, T, L, n_layers = [int(v) for v in g['cfg']]
    seeds = [int(v) for v in g['seeds']]
    ih, iw, sizes = S.PYRAMIDS[str(g['pyramid'])]
    params = {k: v.clone().requires_grad_(True) for k, v in S.make_params(seeds[0], embed_dims=256, num_frames=T, num_points=4, num_levels=L).items()}
    feats = [f.requires_grad_(True) for f in S.make_features(B, T, sizes, seed=seeds[2])]
    metas = S.make_img_metas(B, T, ih, iw)
    for b, m in enumerate(metas):
        m['img_timestamp'] = [float(v) for v in g['timestamps'][b]]
    bbox, feat = g['query_bbox'].clone().requires_grad_(True), g['query_feat'].clone().requires_grad_(True)
    with torch.enable_grad():
        cls, box, _ = O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=n_layers)
        ((cls * g['cot_cls']).sum() + (box * g['cot_box']).sum()).backward()

    def rel(a, b):
        return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()

    errs = {'query_feat': rel(feat.grad, g['grad_query_feat']), 'query_bbox': rel(bbox.grad, g['grad_query_bbox'])}
    for name, gr in list((k, v.grad) for k, v in params.items()) + [('feat%d' % i, f.grad) for i, f in enumerate(feats)]:
        idx = S.grad_sample_indices(gr.numel())
        have = gr.reshape(g['g.' + name].shape) if idx is None else gr.reshape(-1)[idx]
        errs[name] = rel(have, g['g.' + name])
    assert max(errs.values()) < tol, sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    assert bbox.grad[..., 8:].abs().max() == 0
