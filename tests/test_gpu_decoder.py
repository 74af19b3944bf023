"""GPU parity of the drop-in SparseBEVTransformer (sparsebev_amd.transformer) against golden vectors
recorded from the reference decoder, teacher-forced per layer (1e-4) and free-running (sanity bound)."""
import copy

import numpy as np
import pytest
import torch

from conftest import load_golden, runtime_op_by_op, free_running_bound, report_free_running
from sparsebev_amd import synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer, FeaturePyramid, DecoderContext

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = 'cuda:0'
PREFIX = 'decoder.decoder_layer.'


def build(T, L, seed):
    params = S.make_params(seed, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=6, num_levels=L, num_classes=10,
                             code_size=10, pc_range=S.PC_RANGE)
    missing = m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)   # identical key set
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(DEV).eval()


@pytest.mark.parametrize('tag', ['c1', 'c2small', 'L5'])
def test_g7_decoder_teacher_forced_and_free_running(tag):
    g = load_golden('g7_decoder_' + tag)
    B, Q, T, L = [int(v) for v in g['cfg']]
    seeds = [int(v) for v in g['seeds']]
    ih, iw, sizes = S.PYRAMIDS[str(g['pyramid'])]
    model = build(T, L, seeds[0])
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=seeds[2])]
    metas = S.make_img_metas(B, T, ih, iw)
    for b, m in enumerate(metas):
        m['img_timestamp'] = [float(v) for v in g['timestamps'][b]]
    # free-running, through the public forward (the SparseBEVHead call site, sparsebev_head.py:77-83)
    metas_in = copy.deepcopy(metas)
    cls, box = model(g['query_bbox'].to(DEV), g['query_feat'].to(DEV), list(feats), None, metas_in)
    assert cls.shape == g['out_cls'].shape and box.shape == g['out_bbox'].shape
    assert (cls[0].cpu() - g['out_cls'][0]).abs().max() < TOL
    assert (box[0].cpu() - g['out_bbox'][0]).abs().max() < TOL
    # free-running drift, layer by layer, against the reference's OWN drift on these inputs (fixture G13: its two samplers against
    # each other and a one-ulp nudge of query_feat; VERDICT r4 item 4 -- a bare 0.2 stood here): HIP-vs-reference <= 2 x that
    report_free_running('hip', tag, (cls, box), (g['out_cls'], g['out_bbox']), free_running_bound(tag))
    assert 'time_diff' not in metas_in[0] and not torch.is_tensor(metas_in[0]['lidar2img'])   # inputs not mutated
    # the C++ runtime (one call, all layers) and the layer-by-layer Python path launch the same kernels
    cls_lw, box_lw = model(g['query_bbox'].to(DEV), g['query_feat'].to(DEV), list(feats), None, copy.deepcopy(metas), layerwise=True)
    cls_rt, box_rt = runtime_op_by_op(model, g['query_bbox'].to(DEV), g['query_feat'].to(DEV), list(feats), None, copy.deepcopy(metas), exact_gemm=True)
    assert torch.equal(cls_rt, cls_lw) and torch.equal(box_rt, box_lw)
    assert (cls[0] - cls_rt[0]).abs().max() < 2e-5 and (box[0] - box_rt[0]).abs().max() < 2e-5     # row chains: round-off only
    # teacher-forced: each layer from the reference's own inputs
    layer = model.decoder.decoder_layer
    pyr, ctx = FeaturePyramid(feats), DecoderContext(metas, B, torch.device(DEV))
    n = g['out_cls'].shape[0]
    ins = [(g['query_bbox'], g['query_feat'])] + [(g['out_bbox'][i - 1], g['out_feat'][i - 1]) for i in range(1, n)]
    with torch.no_grad():
        for i, (qb, qf) in enumerate(ins):
            x, c, bb = layer(qb.to(DEV), qf.to(DEV), pyr, None, ctx)
            assert (x.cpu() - g['out_feat'][i]).abs().max() < TOL, i
            assert (c.cpu() - g['out_cls'][i]).abs().max() < TOL, i
            assert (bb.cpu() - g['out_bbox'][i]).abs().max() < TOL, i


@pytest.mark.parametrize('mode', ['bf16x3', 'f32', 'bf16x6', 'f16x4'])
@pytest.mark.parametrize('tag', ['c2small'])
def test_g7_decoder_every_gemm_mode_stays_inside_the_parity_budget(tag, mode):
    """The non-default modes of the two big mixing GEMMs (the default, f16x3, is what every other test here runs): teacher-forced
    layers must match the reference recording to 1e-4 in each -- also in the 3 x bf16 split (measured ~1e-5)."""
    g = load_golden('g7_decoder_' + tag)
    B, Q, T, L = [int(v) for v in g['cfg']]
    seeds = [int(v) for v in g['seeds']]
    ih, iw, sizes = S.PYRAMIDS[str(g['pyramid'])]
    model = build(T, L, seeds[0])
    model.decoder.gemm_mode = mode
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=seeds[2])]
    metas = S.make_img_metas(B, T, ih, iw)
    for b, m in enumerate(metas):
        m['img_timestamp'] = [float(v) for v in g['timestamps'][b]]
    n = g['out_cls'].shape[0]
    ins = [(g['query_bbox'], g['query_feat'])] + [(g['out_bbox'][i - 1], g['out_feat'][i - 1]) for i in range(1, n)]
    model.decoder.num_layers = 1
    worst = 0.0
    for i, (qb, qf) in enumerate(ins):
        cls, box = model(qb.to(DEV), qf.to(DEV), list(feats), None, copy.deepcopy(metas))
        worst = max(worst, (cls[0].cpu() - g['out_cls'][i]).abs().max().item(), (box[0].cpu() - g['out_bbox'][i]).abs().max().item())
    assert worst < TOL, worst


def test_g5_self_attention_with_dn_mask():
    g = load_golden('g5_selfattn_T2')
    model = build(2, 4, int(g['seeds'][0]))
    sa = model.decoder.decoder_layer.self_attn
    with torch.no_grad():
        a = sa(g['query_bbox'].to(DEV), g['query_feat'].to(DEV), None)
        b = sa(g['query_bbox'].to(DEV), g['query_feat'].to(DEV), g['mask'].bool().to(DEV))
    assert (a.cpu() - g['out_nomask']).abs().max() < TOL
    assert (b.cpu() - g['out_mask']).abs().max() < TOL


@pytest.mark.parametrize('name,T', [('g4_mixing_T2', 2), ('g4_mixing_T8', 8)])
def test_g4_adaptive_mixing(name, T):
    g = load_golden(name)
    model = build(T, 4, int(g['seeds'][0]))
    with torch.no_grad():
        out = model.decoder.decoder_layer.mixing(g['x'].to(DEV), g['query_feat'].to(DEV))
    assert (out.cpu() - g['out']).abs().max() < TOL


def test_dump_taps_match_reference_recording(tmp_path):
    """A4: with DUMP.enabled the decoder writes the same per-stage files the reference does (viz_sample_points.py
    reads them); the projected coordinates / hit mask must equal the oracle's bit for bit."""
    import os
    from oracle import sparsebev_oracle as O
    from sparsebev_amd.utils import DUMP
    B, Q, T, L = 1, 36, 2, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, L, 21)
    model.decoder.num_layers = 2
    bbox, feat = S.make_queries(B, Q, seed=22)
    feats = S.make_features(B, T, sizes, seed=23)
    metas = S.make_img_metas(B, T, ih, iw)
    DUMP.enabled, DUMP.out_dir = True, str(tmp_path)
    try:
        cls, box = model(bbox.to(DEV), feat.to(DEV), [f.to(DEV) for f in feats], None, copy.deepcopy(metas))
    finally:
        DUMP.enabled = False
    names = ['sample_points_cam', 'sample_points_cam_valid_mask', 'sasa_tau', 'query_bbox', 'bbox_pred', 'cls_score']
    for stage in range(2):
        for n in names:
            assert os.path.exists(os.path.join(str(tmp_path), '%s_stage%d.pth' % (n, stage))), (n, stage)
    uvh = torch.load(os.path.join(str(tmp_path), 'sample_points_cam_stage0.pth'))
    valid = torch.load(os.path.join(str(tmp_path), 'sample_points_cam_valid_mask_stage0.pth'))
    assert uvh.shape == (B, T, 6, Q, 16, 3) and valid.shape == (B, T, 6, Q, 16) and valid.dtype == torch.float32
    # stage 0 inputs are known exactly -> recompute the tap with the oracle from the same sample points
    params = S.make_params(21, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    td = O.time_diff_from_metas(metas, B)
    taps = []
    O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=1, taps=taps)
    # The sample points come out of device expf / sinf / cosf / atan2f, ulps away from the CPU's libm, so the decoder-level mask can
    # differ from the oracle's ONLY where a point projects onto a decision boundary of the hit test (image border u, v in {0, 1},
    # depth homo == eps); the projection itself is bit-exact on identical points (test_gpu_sampling.py).  Assert exactly that:
    # every differing point lies within 1e-4 (normalised image units / metres) of a boundary, and there are at most 3 of them.
    ref_valid, ref_uvh = taps[0]['valid'], taps[0]['uvh']
    diff = valid != ref_valid
    n_diff = int(diff.sum())
    if n_diff:
        u, v, h = ref_uvh[..., 0], ref_uvh[..., 1], ref_uvh[..., 2]
        margin = torch.stack([u.abs(), (u - 1).abs(), v.abs(), (v - 1).abs(), (h - 1e-5).abs()]).min(0).values
        assert margin[diff].max().item() < 1e-4, (n_diff, margin[diff].max().item())
    assert n_diff <= 3, n_diff
    tau = torch.load(os.path.join(str(tmp_path), 'sasa_tau_stage0.pth'))
    assert tau.shape == (B, Q, 8)
    # the dump path (layer-by-layer) and the runtime path give the same numbers
    cls_rt, box_rt = runtime_op_by_op(model, bbox.to(DEV), feat.to(DEV), [f.to(DEV) for f in feats], None, copy.deepcopy(metas), exact_gemm=True)
    assert torch.equal(cls, cls_rt) and torch.equal(box, box_rt)


def test_online_frame_ring_equals_dense_features():
    """pushing frames one at a time into the per-frame ring (only the new frame is relayouted) gives
    bit-identical decoder output to handing the whole [B, T*6, C, H, W] stack over, also after evictions wrapped
    the ring."""
    from sparsebev_amd.cache import FrameFeatureCache
    B, Q, T, L = 2, 36, 4, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, L, 31)
    bbox, feat = S.make_queries(B, Q, seed=32)
    metas = S.make_img_metas(B, T, ih, iw)
    g = torch.Generator(device=DEV).manual_seed(33)
    frames = [[torch.randn(B, 6, 256, h, w, generator=g, device=DEV) for h, w in sizes] for _ in range(T + 3)]   # oldest first
    cache = FrameFeatureCache(T, n_slots=T + 1)
    for i, fr in enumerate(frames):
        cache.push(fr)
        if i + 1 < T:
            continue
        newest_first = frames[i::-1][:T]
        dense = [torch.cat([f[l] for f in newest_first], dim=1) for l in range(L)]           # [B, T*6, C, H, W], t = 0 newest
        a = model(bbox.to(DEV), feat.to(DEV), dense, None, copy.deepcopy(metas))
        r = model(bbox.to(DEV), feat.to(DEV), cache.pyramid(), None, copy.deepcopy(metas))
        assert torch.equal(a[0], r[0]) and torch.equal(a[1], r[1]), i
        lw = model(bbox.to(DEV), feat.to(DEV), cache.pyramid(), None, copy.deepcopy(metas), layerwise=True)
        assert torch.equal(runtime_op_by_op(model, bbox.to(DEV), feat.to(DEV), cache.pyramid(), None, copy.deepcopy(metas), exact_gemm=True)[0], lw[0])
    assert sorted(cache.order) == list(range(T + 1))                                          # every slot in use, no growth


@pytest.mark.parametrize('B,Q,T,pyr', [(1, 900, 8, 'tiny'), (2, 400, 8, 'tiny'), (1, 100, 1, 'tiny5')])
def test_one_layer_at_benchmark_query_counts_vs_oracle(B, Q, T, pyr):
    """The BASELINE configs' query counts (900 = 7 x 128 + 4 rows, 2 x 400, 100) and frame counts through every
    kernel of one decoder layer, against the CPU oracle on the same seeded inputs (1e-4)."""
    from oracle import sparsebev_oracle as O
    ih, iw, sizes = S.PYRAMIDS[pyr]
    L = len(sizes)
    model = build(T, L, 41)
    model.decoder.num_layers = 1
    params = S.make_params(41, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    bbox, feat = S.make_queries(B, Q, seed=42)
    feats = S.make_features(B, T, sizes, seed=43)
    metas = S.make_img_metas(B, T, ih, iw)
    cls, box = model(bbox.to(DEV), feat.to(DEV), [f.to(DEV) for f in feats], None, copy.deepcopy(metas))
    cls_r, box_r, _ = O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=1,
                                sampler=O.msmv_sampling_kernel_semantics)
    assert (cls[0].cpu() - cls_r[0]).abs().max() < TOL
    assert (box[0].cpu() - box_r[0]).abs().max() < TOL


def test_dn_attention_mask_through_runtime():
    """Query-denoising attention mask (models/sparsebev_transformer.py:224-225) through the C++ runtime."""
    from oracle import sparsebev_oracle as O
    B, Q, T, L = 1, 64, 2, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, L, 51)
    model.decoder.num_layers = 1
    params = S.make_params(51, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    bbox, feat = S.make_queries(B, Q, seed=52)
    feats = S.make_features(B, T, sizes, seed=53)
    metas = S.make_img_metas(B, T, ih, iw)
    mask = torch.zeros(Q, Q, dtype=torch.bool)
    mask[:20, 20:] = True            # DN groups cannot see the matching part ...
    mask[20:, :20] = True            # ... and vice versa
    cls, box = model(bbox.to(DEV), feat.to(DEV), [f.to(DEV) for f in feats], mask.to(DEV), copy.deepcopy(metas))
    cls_r, box_r, _ = O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=1,
                                sampler=O.msmv_sampling_kernel_semantics, pre_attn_mask=mask)
    assert (cls[0].cpu() - cls_r[0]).abs().max() < TOL
    assert (box[0].cpu() - box_r[0]).abs().max() < TOL
    cls_n, _ = model(bbox.to(DEV), feat.to(DEV), [f.to(DEV) for f in feats], None, copy.deepcopy(metas))
    assert (cls_n - cls).abs().max() > 1e-3


def test_trained_like_norm1_keeps_f16x3_and_an_outlier_gamma_falls_back_to_the_exact_kernels():
    """the generator input's fp16 scale is an a-priori bound from norm1 (no pass over the activations).  A
    trained-like norm1 (gamma log-uniform in [0.01, 30]) stays within the bound's budget and keeps f16x3 -- and stays within 1e-4 of
    the oracle; a checkpoint with an outlier (one gamma 10^5 times the others) would leave the bulk of the channels without their lo
    image: the runtime refuses the split, warns once and runs the exact f32 kernels -- bit-identical to gemm_mode='f32'."""
    import math
    import warnings
    from oracle import sparsebev_oracle as O
    from sparsebev_amd import runtime as RT
    B, Q, T, L = 1, 64, 2, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    bbox, feat = S.make_queries(B, Q, seed=72)
    feats = S.make_features(B, T, sizes, seed=73)
    metas = S.make_img_metas(B, T, ih, iw)
    g = torch.Generator().manual_seed(74)
    params = S.make_params(71, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    params['norm1.weight'] = torch.exp(torch.rand(256, generator=g) * (math.log(30.0) - math.log(0.01)) + math.log(0.01))
    params['norm1.bias'] = torch.randn(256, generator=g)
    # (the wide gamma makes the generator's input 30x larger: keep the dynamic mixing weights at the scale the synthetic model has)
    params['mixing.parameter_generator.weight'] = params['mixing.parameter_generator.weight'] / 10.0

    def model_of(p, mode):
        m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=1, num_levels=L, num_classes=10, code_size=10, pc_range=S.PC_RANGE)
        m.load_state_dict({PREFIX + k: v for k, v in p.items()}, strict=True)
        m = m.to(DEV).eval()
        m.decoder.gemm_mode = mode
        return m
    args = lambda: (bbox.to(DEV), feat.to(DEV), [f.to(DEV) for f in feats], None, copy.deepcopy(metas))
    m16 = model_of(params, 'f16x3')
    cls, box = m16(*args())
    rt = m16.decoder._runtime
    assert rt.mode_eff == RT.GEMM_F16X3 and 6.0 < rt.f16_headroom_log2 <= RT.F16_MAX_HEADROOM_LOG2
    cls_r, box_r, _ = O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=1, sampler=O.msmv_sampling_kernel_semantics)
    assert (cls[0].cpu() - cls_r[0]).abs().max() < TOL and (box[0].cpu() - box_r[0]).abs().max() < TOL
    # the outlier checkpoint
    bad = dict(params)
    bad['norm1.weight'] = params['norm1.weight'].clone()
    bad['norm1.weight'][17] = 3.0e6
    RT.DecoderRuntime._warned_f16_bound = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        mb = model_of(bad, 'f16x3')
        cls_b, box_b = mb(*args())
    assert mb.decoder._runtime.mode_eff == RT.GEMM_F32 and mb.decoder._runtime.f16_headroom_log2 > RT.F16_MAX_HEADROOM_LOG2
    assert any('exact f32' in str(x.message) for x in w)
    cls_f, box_f = model_of(bad, 'f32')(*args())
    assert torch.equal(cls_b, cls_f) and torch.equal(box_b, box_f)


@pytest.mark.parametrize('P,T,L,pyr', [(8, 2, 5, 'tiny5'), (2, 4, 4, 'tiny'), (8, 15, 4, 'tiny')])
def test_other_point_and_frame_counts_vs_oracle(P, T, L, pyr):
    """num_points / num_frames other than the r50 defaults: the reference's eva02 config uses P = 8, T = 15
    (configs/vit_eva02_1600x640_trainval_future.py:54-58), i.e. 120 in-points for adaptive mixing."""
    from oracle import sparsebev_oracle as O
    B, Q = 1, 36
    ih, iw, sizes = S.PYRAMIDS[pyr]
    params = S.make_params(61, embed_dims=256, num_frames=T, num_points=P, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=P, num_layers=1, num_levels=L, pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    bbox, feat = S.make_queries(B, Q, seed=62)
    feats = S.make_features(B, T, sizes, seed=63)
    metas = S.make_img_metas(B, T, ih, iw)
    cls, box = m(bbox.to(DEV), feat.to(DEV), [f.to(DEV) for f in feats], None, copy.deepcopy(metas))
    cls_r, box_r, _ = O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=1, num_points=P,
                                sampler=O.msmv_sampling_kernel_semantics)
    assert (cls[0].cpu() - cls_r[0]).abs().max() < TOL
    assert (box[0].cpu() - box_r[0]).abs().max() < TOL


def test_captured_graph_replays_bit_identically_and_follows_in_place_updates():
    """sbev_decoder_capture / sbev_graph_launch: a hipGraph of one decoder step must reproduce the eager runtime bit
    for bit, and -- reading its inputs through the captured pointers -- follow in-place input updates."""
    from sparsebev_amd.runtime import DecoderRuntime
    B, Q, T, L = 1, 64, 4, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, L, 41)
    metas = S.make_img_metas(B, T, ih, iw)
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=42)]
    pyr, ctx = FeaturePyramid(feats), DecoderContext(metas, B, torch.device(DEV))
    bbox_a, feat_a = [t.to(DEV) for t in S.make_queries(B, Q, seed=43)]
    bbox_b, feat_b = [t.to(DEV) for t in S.make_queries(B, Q, seed=44)]
    eager = DecoderRuntime(model.decoder)
    ref_a = [t.clone() for t in eager.forward(bbox_a, feat_a, pyr, ctx)]
    ref_b = [t.clone() for t in eager.forward(bbox_b, feat_b, pyr, ctx)]
    assert not torch.equal(ref_a[0], ref_b[0])
    qb, qf = bbox_a.clone(), feat_a.clone()
    graph = DecoderRuntime(model.decoder).capture(qb, qf, pyr, ctx)
    assert graph.num_nodes >= 1 + 6 * 6                    # every launch is a node: 6 per layer with the row chains (17 op by op)
    for _ in range(2):
        cls, box = graph.replay()
        assert torch.equal(cls, ref_a[0]) and torch.equal(box, ref_a[1])
    qb.copy_(bbox_b)
    qf.copy_(feat_b)
    cls, box = graph.replay()
    assert torch.equal(cls, ref_b[0]) and torch.equal(box, ref_b[1])
    graph.destroy()
    with pytest.raises(RuntimeError):
        graph.replay()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_bf16_feature_storage_through_the_runtime(dtype):
    """bf16 / fp16 feature maps (storage only: exact widening, fp32 math) handed over channels-last, as configs[4] (ViT neck,
    bf16) or the reference's fp16 eval mode (val.py:115) would: the decoder must equal the oracle run on the widened features."""
    from oracle import sparsebev_oracle as O
    B, Q, T, L = 1, 36, 2, 5
    ih, iw, sizes = S.PYRAMIDS['tiny5']
    params = S.make_params(51, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    model = build(T, L, 51)
    bbox, feat = S.make_queries(B, Q, seed=52)
    metas = S.make_img_metas(B, T, ih, iw)
    feats16 = [f.to(dtype) for f in S.make_features(B, T, sizes, seed=53)]
    dev_feats = [f.to(DEV).permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats16]     # NHWC memory
    cls, box = model(bbox.to(DEV), feat.to(DEV), dev_feats, None, copy.deepcopy(metas))
    ref_cls, ref_box, _ = O.decoder(params, bbox, feat, [f.float() for f in feats16], metas, S.PC_RANGE, num_layers=1)
    assert (cls[0].cpu() - ref_cls[0]).abs().max() < TOL and (box[0].cpu() - ref_box[0]).abs().max() < TOL
    lw = model(bbox.to(DEV), feat.to(DEV), dev_feats, None, copy.deepcopy(metas), layerwise=True)
    assert torch.equal(runtime_op_by_op(model, bbox.to(DEV), feat.to(DEV), dev_feats, None, copy.deepcopy(metas), exact_gemm=True)[0], lw[0])


def test_full_size_config2_properties():
    """BASELINE config 2 at full size (r50 704x256 pyramid, 900 queries, T = 8, 6 layers) -- too big for the CPU oracle in a
    test, so size-independent properties: run-to-run bit determinism (no atomics, no data-dependent order anywhere on
    the forward path), batch consistency (a sample computed alone and as one of two gives the same layer-0 output to
    rounding: other GEMM tilings are picked for other row counts), finite outputs, and insensitivity of layer 0 to the
    order of the queries (every op is per-query except the attention, whose key order only changes rounding)."""
    T, L, Q = 8, 4, 900
    ih, iw, sizes = S.PYRAMIDS['r50_704x256']
    model = build(T, L, 71)
    feats = S.make_features(1, T, sizes, seed=72, device=DEV)
    bbox, feat = [t.to(DEV) for t in S.make_queries(1, Q, seed=73)]
    metas = S.make_img_metas(1, T, ih, iw)
    cls, box = model(bbox, feat, feats, None, copy.deepcopy(metas))
    cls2, box2 = model(bbox, feat, feats, None, copy.deepcopy(metas))
    assert cls.shape == (6, 1, Q, 10) and torch.isfinite(cls).all() and torch.isfinite(box).all()
    assert torch.equal(cls, cls2) and torch.equal(box, box2)
    # the same sample twice in a batch of 2
    feats2 = [f.expand(2, -1, -1, -1, -1).contiguous() for f in feats]
    clsb, boxb = model(bbox.expand(2, -1, -1).contiguous(), feat.expand(2, -1, -1).contiguous(), feats2, None,
                       copy.deepcopy(S.make_img_metas(2, T, ih, iw)))
    assert torch.equal(clsb[:, 0], clsb[:, 1])
    assert (clsb[0, 0] - cls[0, 0]).abs().max() < TOL and (boxb[0, 0] - box[0, 0]).abs().max() < TOL
    # query permutation
    perm = torch.randperm(Q, generator=torch.Generator().manual_seed(74)).to(DEV)
    clsp, boxp = model(bbox[:, perm].contiguous(), feat[:, perm].contiguous(), feats, None, copy.deepcopy(metas))
    assert (clsp[0] - cls[0][:, perm]).abs().max() < TOL and (boxp[0] - box[0][:, perm]).abs().max() < TOL


def test_per_call_constants_survive_the_pinned_upload_ring():
    # DecoderContext stages time_diff / lidar2img through a ring of page-locked buffers (asynchronous upload): more
    # contexts than ring slots, built back to back while the device is busy, must each keep their own values
    from sparsebev_amd.transformer import _PinnedUpload
    B, T = 1, 8
    busy = torch.randn(4096, 4096, device=DEV)
    ctxs, want = [], []
    for i in range(3 * _PinnedUpload.DEPTH):
        metas = S.make_img_metas(B, T, 256, 704)
        for m in metas:
            m['lidar2img'] = [a + np.float32(i) for a in m['lidar2img']]      # a different set of matrices per context
        for _ in range(4):
            busy = busy @ busy * 1e-3                      # keep the stream ahead of the uploads
        ctxs.append(DecoderContext(metas, B, torch.device(DEV)))
        want.append(np.asarray([m['lidar2img'] for m in metas]).astype(np.float32))
    torch.cuda.synchronize()
    for c, w in zip(ctxs, want):
        assert np.array_equal(c.lidar2img.cpu().numpy(), w)
    assert not np.array_equal(want[0], want[1])


def test_profile_stride_brackets_every_nth_call_only():
    # sbev_profile_stride: the event records around a launch are not free, so bench.py brackets the sampler in every 5th
    # timed step only; here: stride 3 over 7 calls -> calls 0, 3, 6 are bracketed, num_layers sampler launches each
    from sparsebev_amd import runtime as rt
    from sparsebev_amd.runtime import DecoderRuntime
    B, Q, T, L = 1, 64, 4, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, L, 51)
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=52)]
    pyr, ctx = FeaturePyramid(feats), DecoderContext(S.make_img_metas(B, T, ih, iw), B, torch.device(DEV))
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=53)]
    run = DecoderRuntime(model.decoder)
    ref = [t.clone() for t in run.forward(bbox, feat, pyr, ctx)]
    rt.read_kernel_ms(3)                                       # drop anything an earlier test left behind
    rt.read_sampler_ms()
    try:
        rt.profile_stride(3)
        rt.profile_sampler(1 | 8)                              # the stand-alone sampler and the fused gather + mixing launch
        for _ in range(7):
            out = run.forward(bbox, feat, pyr, ctx)
        torch.cuda.synchronize()
        ms = rt.read_sampler_ms() + rt.read_kernel_ms(3)       # this shape (T*P = 16) runs fused: kind 3; kind 0 stays empty
    finally:
        rt.profile_sampler(False)
        rt.profile_stride(1)
    assert len(ms) == 3 * model.decoder.num_layers and all(0 < m < 10 for m in ms)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])       # bracketing changes no result
    with pytest.raises(Exception):
        rt.profile_stride(0)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_two_byte_ring_equals_the_dense_pyramid_of_the_same_frames(dtype):
    """FrameFeatureCache(dtype=fp16 / bf16) (round 4): frames stay in their storage type (NCHW frames through the 2-byte relayout,
    channels-last ones copied), and the decoder on the ring equals the decoder on the dense stack of the same frames -- and the fp32
    decoder on the widened stack -- bit for bit, also after evictions wrapped the ring; other frame types are refused."""
    from sparsebev_amd.cache import FrameFeatureCache
    B, Q, T, L = 2, 36, 4, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, L, 31)
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=32)]
    metas = S.make_img_metas(B, T, ih, iw)
    g = torch.Generator(device=DEV).manual_seed(34)
    frames = [[torch.randn(B, 6, 256, h, w, generator=g, device=DEV).to(dtype) for h, w in sizes] for _ in range(T + 3)]
    cache = FrameFeatureCache(T, n_slots=T + 1, dtype=dtype)
    for i, fr in enumerate(frames):
        if i % 2:                                                 # every other frame arrives channels-last
            fr = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in fr]
        cache.push(fr)
        if i + 1 < T:
            continue
        assert all(b.dtype == dtype for b in cache.buffers)
        newest_first = frames[i::-1][:T]
        dense = [torch.cat([f[l] for f in newest_first], dim=1) for l in range(L)]
        a = model(bbox, feat, dense, None, copy.deepcopy(metas))
        r = model(bbox, feat, cache.pyramid(), None, copy.deepcopy(metas))
        w = model(bbox, feat, [d.float() for d in dense], None, copy.deepcopy(metas))
        assert torch.equal(a[0], r[0]) and torch.equal(a[1], r[1]), i
        assert torch.equal(w[0], r[0]) and torch.equal(w[1], r[1]), i
    with pytest.raises(RuntimeError):
        cache.push([f.float() for f in frames[0]])
    with pytest.raises(ValueError):
        FrameFeatureCache(T, dtype=torch.float64)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_ring_takes_channels_last_frames_through_the_widening_copy(dtype):
    """FrameFeatureCache.push with channels-last memory (what a channels_last conv stack emits, fp32 / fp16 / bf16): the frame is
    copied + widened straight into its slot by sbev_copy_widen_f32 (no relayout, no torch kernel) and must equal the values a
    dense NCHW push of the widened tensor stores."""
    from sparsebev_amd.cache import FrameFeatureCache
    B, T = 2, 2
    ih, iw, sizes = S.PYRAMIDS['tiny']
    g = torch.Generator(device=DEV).manual_seed(7)
    ring_cl, ring_ref = FrameFeatureCache(T), FrameFeatureCache(T)
    for _ in range(3):
        frame = [torch.randn(B, 6, 256, h, w, generator=g, device=DEV).to(dtype) for h, w in sizes]
        cl = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in frame]        # NHWC memory, NCHW axes
        assert cl[0].stride(2) == 1
        ring_cl.push(cl)
        ring_ref.push([f.float().contiguous() for f in frame])
    assert ring_cl.order == ring_ref.order
    for a, b in zip(ring_cl.buffers, ring_ref.buffers):
        assert torch.equal(a, b)
