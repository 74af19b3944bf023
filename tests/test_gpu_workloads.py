"""GPU parity at the real BASELINE.json workload shapes (c2, c3, c4, c5).

Every other decoder test runs on reduced pyramids / query counts so that the CPU oracle finishes instantly; the
kernels, however, pick other tilings, split plans and code paths at the workload sizes (M = B*Q = 3200 / 3600 rows
through the strip / register-tile GEMMs, the LayerNorm-prologue fallback above 2048 rows, 5-level and bf16 sampler
instantiations, > 2^31-byte feature levels).  Here the decoder layer runs at exactly those shapes and is compared with
the oracle on the same seeded inputs (1e-4, the north_star tolerance).  The oracle is evaluated one sample at a time
(samples are independent) so that host memory stays at one sample's features."""
import copy

import numpy as np
import pytest
import torch

from conftest import runtime_op_by_op
from sparsebev_amd import synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = 'cuda:0'
GEMM_MODES_UNDER_TEST = ('f32', 'bf16x6', 'f16x4', 'bf16x3s', 'bf16x3')      # besides the default ('f16x3')
PREFIX = 'decoder.decoder_layer.'
DEFAULT_GEMM = 'f16x3'

# name: (pyramid, Q, T, per-GPU batch, feature dtype) -- bench.py's CONFIGS, BASELINE.json configs[1..4]
WORKLOADS = {
    'c2': ('r50_704x256', 900, 8, 1, torch.float32),
    'c3': ('r50_704x256', 400, 8, 8, torch.float32),
    'c4': ('r101_1408x512', 900, 8, 4, torch.float32),
    'c5': ('eva02_1600x640', 900, 8, 1, torch.bfloat16),
    # the reference's own largest config (configs/vit_eva02_1600x640_trainval_future.py:54-58): 15 frames, 8 points, 1600 queries
    'c6': ('eva02_1600x640', 1600, 15, 1, torch.bfloat16, 8),
}


def build(T, L, seed, num_layers=1, P=4):
    params = S.make_params(seed, embed_dims=256, num_frames=T, num_points=P, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=P, num_layers=num_layers, num_levels=L, num_classes=10,
                             code_size=10, pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    return m.to(DEV).eval(), params


def workload(name, seed):
    pyr, Q, T, B, dt = WORKLOADS[name][:5]
    ih, iw, sizes = S.PYRAMIDS[pyr]
    feats = S.make_features(B, T, sizes, seed=seed, device=DEV, dtype=dt)        # generated on the device, left resident
    if dt != torch.float32:                                                      # a bf16 neck hands channels-last memory over
        feats = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]
    bbox, feat = S.make_queries(B, Q, seed=seed + 1)
    metas = S.make_img_metas(B, T, ih, iw)
    return feats, bbox, feat, metas, (B, Q, T, len(sizes))


def points_of(name):
    return WORKLOADS[name][5] if len(WORKLOADS[name]) > 5 else 4


def oracle_per_sample(params, bbox, feat, feats_dev, metas, num_layers=1, forced=None, P=4, sampler=None):
    """O.decoder one sample at a time on the widened-to-fp32 CPU copy of that sample's features (kernel-semantics sampler)."""
    from oracle import sparsebev_oracle as O
    cls, box, x = [], [], []
    for b in range(bbox.shape[0]):
        fb = [f[b:b + 1].to(next(iter(params.values())).dtype).cpu().contiguous() for f in feats_dev]      # (fp32, or fp64 for the exact evaluation)
        fi = None if forced is None else [(qb[b:b + 1], qf[b:b + 1]) for qb, qf in forced]
        c, bb, xx = O.decoder(params, bbox[b:b + 1], feat[b:b + 1], fb, metas[b:b + 1], S.PC_RANGE, num_layers=num_layers,
                              num_points=P, sampler=sampler or O.msmv_sampling_kernel_semantics, forced_inputs=fi)
        cls.append(c), box.append(bb), x.append(xx)
        del fb
    return torch.cat(cls, 1), torch.cat(box, 1), torch.cat(x, 1)


@pytest.mark.parametrize('name', ['c2', 'c3', 'c4', 'c5', 'c6'])
def test_one_layer_at_full_workload_shape_vs_oracle(name):
    """One decoder layer at the full workload shape (all B samples, the real pyramid), C++ runtime and layer-by-layer
    path, against the oracle: cls / bbox / query_feat to 1e-4."""
    feats, bbox, feat, metas, (B, Q, T, L) = workload(name, seed=101)
    model, params = build(T, L, seed=100, P=points_of(name))
    cls, box = model(bbox.to(DEV), feat.to(DEV), list(feats), None, copy.deepcopy(metas))
    assert cls.shape == (1, B, Q, 10) and box.shape == (1, B, Q, 10)
    lw_cls, lw_box = model(bbox.to(DEV), feat.to(DEV), list(feats), None, copy.deepcopy(metas), layerwise=True)
    rt_cls, rt_box = runtime_op_by_op(model, bbox.to(DEV), feat.to(DEV), list(feats), None, copy.deepcopy(metas), exact_gemm=True)
    assert torch.equal(rt_cls, lw_cls) and torch.equal(rt_box, lw_box)
    ref_cls, ref_box, _ = oracle_per_sample(params, bbox, feat, feats, metas, P=points_of(name))
    assert (cls.cpu() - ref_cls).abs().max() < TOL                   # (c2 / c5: through the row-chain kernels)
    assert (box.cpu() - ref_box).abs().max() < TOL
    assert (lw_cls.cpu() - ref_cls).abs().max() < TOL and (lw_box.cpu() - ref_box).abs().max() < TOL
    # the same assertion in every GEMM mode bench.py prints: split-bf16 parameter generator + out-projection
    for mode in GEMM_MODES_UNDER_TEST:
        model.decoder.gemm_mode = mode
        try:
            m_cls, m_box = model(bbox.to(DEV), feat.to(DEV), list(feats), None, copy.deepcopy(metas))
            o_cls, o_box = runtime_op_by_op(model, bbox.to(DEV), feat.to(DEV), list(feats), None, copy.deepcopy(metas))
        finally:
            model.decoder.gemm_mode = DEFAULT_GEMM
        for got, want in ((m_cls, ref_cls), (m_box, ref_box), (o_cls, ref_cls), (o_box, ref_box)):
            err = (got.cpu() - want).abs().max().item()
            assert err < TOL, (name, mode, err)


def test_c2_six_layers_teacher_forced_vs_oracle():
    """BASELINE config 2 at full size, all six layers: the oracle runs free (6 layers), and every layer of the HIP path
    starts from the oracle's own (bbox, feat) of the previous layer -- rounding noise grows ~5x per random-init layer when
    free-running, so teacher forcing is what makes a 1e-4 assertion meaningful."""
    from sparsebev_amd.transformer import FeaturePyramid, DecoderContext
    feats, bbox, feat, metas, (B, Q, T, L) = workload('c2', seed=111)
    model, params = build(T, L, seed=110)
    ref_cls, ref_box, ref_x = oracle_per_sample(params, bbox, feat, feats, metas, num_layers=6)
    ins = [(bbox, feat)] + [(ref_box[i - 1], ref_x[i - 1]) for i in range(1, 6)]
    layer = model.decoder.decoder_layer
    pyr, ctx = FeaturePyramid(feats), DecoderContext(metas, B, torch.device(DEV))
    worst = 0.0
    with torch.no_grad():
        for i, (qb, qf) in enumerate(ins):
            cls, box = model(qb.to(DEV), qf.to(DEV), pyr, None, copy.deepcopy(metas))            # C++ runtime, 1 layer (row chains)
            rt = runtime_op_by_op(model, qb.to(DEV), qf.to(DEV), pyr, None, copy.deepcopy(metas), exact_gemm=True)
            x, c, bb = layer(qb.to(DEV), qf.to(DEV), pyr, None, ctx)                            # same kernels, op by op
            assert torch.equal(rt[0][0], c) and torch.equal(rt[1][0], bb)
            for got, want in ((x, ref_x[i]), (c, ref_cls[i]), (bb, ref_box[i]), (cls[0], ref_cls[i]), (box[0], ref_box[i])):
                err = (got.cpu() - want).abs().max().item()
                worst = max(worst, err)
                assert err < TOL, (i, err)
            for mode in GEMM_MODES_UNDER_TEST:                 # every layer, every GEMM mode, same 1e-4
                model.decoder.gemm_mode = mode
                try:
                    m_cls, m_box = model(qb.to(DEV), qf.to(DEV), pyr, None, copy.deepcopy(metas))
                finally:
                    model.decoder.gemm_mode = DEFAULT_GEMM
                for got, want in ((m_cls[0], ref_cls[i]), (m_box[0], ref_box[i])):
                    err = (got.cpu() - want).abs().max().item()
                    assert err < TOL, (i, mode, err)
    # the free-running 6-layer forward: finite, its first layer is the teacher-forced one, and its drift from the oracle's free run is
    # held against the drift the ARITHMETIC shows against itself at this very shape (the G13 yardstick, fixture-pinned at c1 / c2small,
    # rebuilt here at full c2 from the pinned oracle: its two samplers against each other, one-ulp nudges of query_feat, and its fp32
    # evaluation against its fp64 one): per layer <= 2 x that, as long as the yardstick itself is below saturation
    from oracle import sparsebev_oracle as O
    model.decoder.num_layers = 6
    cls6, box6 = model(bbox.to(DEV), feat.to(DEV), pyr, None, copy.deepcopy(metas))
    assert torch.isfinite(cls6).all() and torch.isfinite(box6).all()
    assert (cls6[0].cpu() - ref_cls[0]).abs().max() < TOL
    runs = [oracle_per_sample(params, bbox, feat, feats, metas, num_layers=6, sampler=O.msmv_sampling_gridsample)]      # the other sampler
    for off, way in ((0, 1.0), (1, -1.0)):                                                                                # one-ulp nudges
        nudged = feat.clone()
        nudged.view(-1)[off::2] = torch.nextafter(nudged.view(-1)[off::2], torch.full_like(nudged.view(-1)[off::2], way * float('inf')))
        runs.append(oracle_per_sample(params, bbox, nudged, feats, metas, num_layers=6))
    # ... and the same arithmetic evaluated in fp64: how far the fp32 evaluation itself sits from the exact one (the yardstick that moves
    # EVERY op's rounding, as an independent implementation does -- the first two only move the sampler's / the input's)
    p64 = {k: v.double() for k, v in params.items()}
    runs.append(oracle_per_sample(p64, bbox.double(), feat.double(), [f.double() for f in feats], metas, num_layers=6))

    def div(a, b):
        return np.array([(a[i].cpu().double() - b[i].cpu().double()).abs().max().item() for i in range(6)])
    for what, got, k in (('cls', cls6, 0), ('bbox', box6, 1)):
        ref = (ref_cls, ref_box)[k]
        yard = np.max(np.stack([div(ref, r[k]) for r in runs]), axis=0)
        d = div(got, ref)
        print('free-running hip c2 (full shape) %-4s divergence per layer %s | oracle-vs-itself %s'
              % (what, ' '.join('%.1e' % v for v in d), ' '.join('%.1e' % v for v in yard)))
        live = yard < 0.05                 # beyond that two runs of the SAME arithmetic are decorrelated (O(1) apart): nothing left to compare
        assert live[:2].all() and (d[live] <= 2.0 * np.maximum(yard[live], 2e-6)).all(), (what, d, yard)


@pytest.mark.parametrize('name', ['c3', 'c4', 'c5', 'c6'])
def test_full_workload_six_layer_properties(name):
    """Six layers at the full c3 / c4 / c5 shapes: size-independent properties (the 6-layer oracle at these sizes is
    minutes of CPU): bit determinism run to run, the C++ runtime equals the layer-by-layer path bit for bit, finite
    outputs, and a sample computed alone equals the same sample inside the batch to rounding."""
    feats, bbox, feat, metas, (B, Q, T, L) = workload(name, seed=121)
    model, _ = build(T, L, seed=120, num_layers=6, P=points_of(name))
    qb, qf = bbox.to(DEV), feat.to(DEV)
    cls, box = model(qb, qf, list(feats), None, copy.deepcopy(metas))
    cls2, box2 = model(qb, qf, list(feats), None, copy.deepcopy(metas))
    assert cls.shape == (6, B, Q, 10) and torch.isfinite(cls).all() and torch.isfinite(box).all()
    assert torch.equal(cls, cls2) and torch.equal(box, box2)
    lw_cls, lw_box = model(qb, qf, list(feats), None, copy.deepcopy(metas), layerwise=True)
    rt_cls, rt_box = runtime_op_by_op(model, qb, qf, list(feats), None, copy.deepcopy(metas), exact_gemm=True)
    assert torch.equal(rt_cls, lw_cls) and torch.equal(rt_box, lw_box)
    if B > 1:
        b = B - 1
        one = model(qb[b:b + 1].contiguous(), qf[b:b + 1].contiguous(), [f[b:b + 1].contiguous() for f in feats], None,
                    copy.deepcopy(metas[b:b + 1]))
        assert (one[0][0, 0] - cls[0, b]).abs().max() < TOL and (one[1][0, 0] - box[0, b]).abs().max() < TOL
