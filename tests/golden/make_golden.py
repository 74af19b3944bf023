"""Generate the golden fixtures in this directory by running the REFERENCE's own code.

Runs only in the authoring container (needs /root/reference; never on the GPU box).  The reference
package is imported without executing its mmcv/mmdet-dependent ``__init__`` files (stub packages with
``__path__``) and with five tiny stand-ins for the mmcv/mmdet symbols ``models/sparsebev_transformer.py``
imports.  The mmcv stand-ins restate mmcv-full 1.6.0 semantics
(``MultiheadAttention`` = identity + torch.nn.MultiheadAttention, ``FFN`` = identity + Linear-ReLU-Linear);
mmcv itself is not vendored by the reference, so parity is unpinned AT THAT BOUNDARY ONLY and the
underlying math (torch.nn.MultiheadAttention / nn.Linear) is the oracle there.

Inputs too large to commit (decoder weights, feature pyramids) are regenerated from seeds by
``sparsebev_amd.synthetic``; each fixture stores a checksum of every regenerated tensor.

    python tests/golden/make_golden.py
"""
import importlib
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from sparsebev_amd import synthetic as S   # noqa: E402

REF = '/root/reference'
PREFIX = 'decoder.decoder_layer.'


def _stub(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path:
        m.__path__ = [path]
    m.__dict__.update(attrs)
    sys.modules[name] = m


def import_reference():
    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()

    class MultiheadAttention(BaseModule):      # mmcv 1.6.0 cnn/bricks/transformer.py semantics
        def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., batch_first=False):
            super().__init__()
            self.batch_first = batch_first
            self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop)

        def forward(self, query, attn_mask=None):
            q = query.transpose(0, 1) if self.batch_first else query
            out = self.attn(q, q, q, attn_mask=attn_mask)[0]
            return query + (out.transpose(0, 1) if self.batch_first else out)

    class FFN(BaseModule):
        def __init__(self, embed_dims, feedforward_channels, ffn_drop=0.):
            super().__init__()
            self.layers = nn.Sequential(
                nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(ffn_drop)),
                nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))

        def forward(self, x):
            return x + self.layers(x)

    class _Registry:
        def register_module(self):
            return lambda cls: cls

    _stub('models', REF + '/models'); _stub('models.bbox', REF + '/models/bbox'); _stub('models.csrc', REF + '/models/csrc')
    _stub('mmcv'); _stub('mmcv.runner', BaseModule=BaseModule)
    _stub('mmcv.cnn', bias_init_with_prob=lambda p: float(-math.log((1 - p) / p)))
    _stub('mmcv.cnn.bricks'); _stub('mmcv.cnn.bricks.transformer', MultiheadAttention=MultiheadAttention, FFN=FFN)
    _stub('mmdet'); _stub('mmdet.models'); _stub('mmdet.models.utils'); _stub('mmdet.models.utils.builder', TRANSFORMER=_Registry())
    tr = importlib.import_module('models.sparsebev_transformer')
    smp = importlib.import_module('models.sparsebev_sampling')
    wrap = importlib.import_module('models.csrc.wrapper')
    utils = importlib.import_module('models.utils')
    return tr, smp, wrap, utils


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


def cf_from_cl(feats_cl):
    return [f.permute(0, 4, 1, 2, 3).contiguous() for f in feats_cl]


def edge_locs(Bp, Q, P, level_sizes, g):
    """Coordinates covering interior points, exact 0/1, just outside (within one pixel), far outside."""
    loc = torch.rand(Bp, Q, P, 3, generator=g)
    loc[..., 2] = torch.randint(0, 6, (Bp, Q, P), generator=g).float() / 5
    H0, W0 = level_sizes[0]
    specials = [0.0, 1.0, -0.5 / (W0 - 1), 1 + 0.5 / (W0 - 1), -0.5 / (H0 - 1), 1 + 0.99 / (H0 - 1),
                -1.0 / (W0 - 1), 1 + 1.0 / (W0 - 1), -3.0, 4.0, 0.5, 1e-7, 1 - 1e-7]
    flat = loc.view(-1, 3)
    for i, s in enumerate(specials):
        flat[3 * i, 0] = s
        flat[3 * i + 1, 1] = s
        flat[3 * i + 2, 0] = s
        flat[3 * i + 2, 1] = specials[(i + 5) % len(specials)]
    return loc


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tr, smp, wrap, utils = import_reference()
    assert wrap.MSMV_CUDA is False

    # ---- G1: msmv_sampling_pytorch (csrc/wrapper.py:14-38) ------------------------------------------------
    for tag, pyr, C, Bp in (('L4_C8', 'tiny', 8, 8), ('L4_C64', 'tiny', 64, 4), ('L5_C8', 'tiny5', 8, 4), ('L5_C64', 'tiny5', 64, 1)):
        g = torch.Generator().manual_seed(100 + C + len(S.PYRAMIDS[pyr][2]))
        sizes = S.PYRAMIDS[pyr][2]
        Q, P, L = 16, 4, len(sizes)
        feats_cl = [torch.randn(Bp, 6, h, w, C, generator=g) for h, w in sizes]
        loc = edge_locs(Bp, Q, P, sizes, g)
        wts = torch.softmax(torch.randn(Bp, Q, P, L, generator=g), -1)
        out = wrap.msmv_sampling_pytorch(cf_from_cl(feats_cl), loc, wts)
        save('g1_msmv_' + tag, loc=loc, weights=wts, out=out, sizes=np.array(sizes),
             **{'feat%d' % i: f for i, f in enumerate(feats_cl)})
    # P = 7 (odd point count, exercises the generic-P path of the HIP kernel)
    g = torch.Generator().manual_seed(7)
    sizes = S.PYRAMIDS['tiny'][2]
    feats_cl = [torch.randn(3, 6, h, w, 16, generator=g) for h, w in sizes]
    loc = edge_locs(3, 9, 7, sizes, g)
    wts = torch.softmax(torch.randn(3, 9, 7, 4, generator=g), -1)
    out = wrap.msmv_sampling_pytorch(cf_from_cl(feats_cl), loc, wts)
    save('g1_msmv_L4_C16_P7', loc=loc, weights=wts, out=out, sizes=np.array(sizes),
         **{'feat%d' % i: f for i, f in enumerate(feats_cl)})

    # ---- G8: backward of msmv_sampling_pytorch by autograd (the reference's training path without the CUDA op) ----
    for tag, pyr, C, Bp in (('L4_C8', 'tiny', 8, 3), ('L5_C64', 'tiny5', 64, 1)):
        g = torch.Generator().manual_seed(800 + C)
        sizes = S.PYRAMIDS[pyr][2]
        Q, P, L = 12, 4, len(sizes)
        feats_cl = [torch.randn(Bp, 6, h, w, C, generator=g) for h, w in sizes]
        loc = torch.rand(Bp, Q, P, 3, generator=g) * 1.2 - 0.1          # interior + a border band on every side
        loc[..., 2] = torch.randint(0, 6, (Bp, Q, P), generator=g).float() / 5
        wts = torch.softmax(torch.randn(Bp, Q, P, L, generator=g), -1)
        gout = torch.randn(Bp, Q, C, P, generator=g)
        feats_cf = [f.clone().requires_grad_(True) for f in cf_from_cl(feats_cl)]
        loc_g, wts_g = loc.clone().requires_grad_(True), wts.clone().requires_grad_(True)
        with torch.enable_grad():
            out = wrap.msmv_sampling_pytorch(feats_cf, loc_g, wts_g)
            out.backward(gout)
        save('g8_msmv_bwd_' + tag, loc=loc, weights=wts, grad_out=gout, sizes=np.array(sizes),
             grad_loc_xy=loc_g.grad[..., :2], grad_weights=wts_g.grad,
             **{'feat%d' % i: f for i, f in enumerate(feats_cl)},
             **{'grad_feat%d' % i: f.grad.permute(0, 2, 3, 4, 1).contiguous() for i, f in enumerate(feats_cf)})

    # ---- G2: sampling_4d with the DUMP taps (sparsebev_sampling.py:27-130) --------------------------------
    for T, Q in ((1, 40), (8, 25)):
        g = torch.Generator().manual_seed(200 + T)
        B, G, P, C = (2 if T == 1 else 1), 4, 4, 8
        ih, iw, sizes = S.PYRAMIDS['tiny']
        L = len(sizes)
        metas = S.make_img_metas(B, T, ih, iw)
        lidar2img = torch.from_numpy(np.asarray([m['lidar2img'] for m in metas]).astype(np.float32))
        # points on a ring of 5..45 m radius around the ego, heights -3..2 m: 0-, 1- and 2-hit cases all occur
        r = 5 + 40 * torch.rand(B, Q, T, G, P, generator=g)
        a = 2 * math.pi * torch.rand(B, Q, T, G, P, generator=g)
        pts = torch.stack([r * torch.cos(a), r * torch.sin(a), -3 + 5 * torch.rand(B, Q, T, G, P, generator=g)], -1)
        sw = torch.softmax(torch.randn(B, Q, G, T, P, L, generator=g), -1)       # NOT T-expanded: pins quirk q1
        feats = [torch.randn(B, T * 6, G * C, h, w, generator=g) for h, w in sizes]
        feats_cf = [f.reshape(B, T, 6, G, C, *f.shape[-2:]).permute(0, 1, 3, 4, 2, 5, 6).reshape(B * T * G, C, 6, *f.shape[-2:]).contiguous()
                    for f in feats]
        utils.DUMP.enabled = True
        utils.DUMP.stage_count = 0
        out = smp.sampling_4d(pts, feats_cf, sw, lidar2img, ih, iw)
        utils.DUMP.enabled = False
        uvh = torch.load('%s/sample_points_cam_stage0.pth' % utils.DUMP.out_dir)
        valid = torch.load('%s/sample_points_cam_valid_mask_stage0.pth' % utils.DUMP.out_dir)
        nh = valid.sum(2)
        print('  G2 T=%d hits: none %.3f one %.3f two+ %.3f' % (T, (nh == 0).float().mean(), (nh == 1).float().mean(), (nh >= 2).float().mean()))
        save('g2_sampling4d_T%d' % T, sample_points=pts, scale_weights=sw, lidar2img=lidar2img, image_hw=np.array([ih, iw]),
             uvh=uvh, valid=valid.to(torch.uint8), out=out, sizes=np.array(sizes),
             **{'feat%d' % i: f for i, f in enumerate(feats)})

    # ---- shared decoder config for G3..G7 -----------------------------------------------------------------
    def build(T, L, seed):
        cfg = dict(embed_dims=256, num_frames=T, num_points=4, num_levels=L)
        params = S.make_params(seed, **cfg)
        m = tr.SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=6, num_levels=L, num_classes=10,
                                    code_size=10, pc_range=S.PC_RANGE)
        m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
        return m.eval(), params

    def metas_for(B, T, pyr):
        ih, iw, sizes = S.PYRAMIDS[pyr]
        metas = S.make_img_metas(B, T, ih, iw)
        for b, m in enumerate(metas):     # non-uniform per-camera timestamps so time_diff is not a multiple of 0.5
            m['img_timestamp'] = [ts + 0.003 * ((i * 7 + b) % 6) for i, ts in enumerate(m['img_timestamp'])]
        return metas, ih, iw, sizes

    with torch.no_grad():
        # ---- G3 / G4 / G5 / G6 at T=2 (small) and G4 also at T=8 ------------------------------------------
        B, Q, T, L = 2, 36, 2, 4
        model, params = build(T, L, seed=3)
        layer = model.decoder.decoder_layer
        metas, ih, iw, sizes = metas_for(B, T, 'tiny')
        bbox, feat = S.make_queries(B, Q, seed=31)
        feats = S.make_features(B, T, sizes, seed=32)
        # run the reference decoder's own preamble (time_diff, lidar2img, feature regroup :60-85)
        import copy
        metas_run = copy.deepcopy(metas)
        ts = np.array([m['img_timestamp'] for m in metas_run], dtype=np.float64).reshape(B, -1, 6)
        metas_run[0]['time_diff'] = torch.from_numpy(np.mean(ts[:, :1] - ts, axis=-1).astype(np.float32))
        metas_run[0]['lidar2img'] = torch.from_numpy(np.asarray([m['lidar2img'] for m in metas]).astype(np.float32))
        feats_cf = []
        for f in feats:
            Bf, TN, GC, H, W = f.shape
            f = f.reshape(Bf, T, 6, 4, GC // 4, H, W).permute(0, 1, 3, 4, 2, 5, 6).reshape(Bf * T * 4, GC // 4, 6, H, W)
            feats_cf.append(f.contiguous())
        common = dict(query_bbox=bbox, query_feat=feat, time_diff=metas_run[0]['time_diff'],
                      lidar2img=metas_run[0]['lidar2img'], image_hw=np.array([ih, iw]), sizes=np.array(sizes),
                      cfg=np.array([B, Q, T, L]), seeds=np.array([3, 31, 32]),
                      params_checksum=S.checksum(params), feats_checksum=S.checksum(feats))
        g3 = layer.sampling.inner_forward(bbox, feat, feats_cf, metas_run)
        save('g3_sampling_T2', out=g3, **common)
        x_mix = torch.randn(B, Q, 4, T * 4, 64, generator=torch.Generator().manual_seed(33))
        save('g4_mixing_T2', x=x_mix, out=layer.mixing.inner_forward(x_mix, feat), **common)
        mask = torch.zeros(Q, Q, dtype=torch.bool)
        mask[:10, 10:] = True
        mask[10:, :4] = True
        save('g5_selfattn_T2', out_nomask=layer.self_attn.inner_forward(bbox, feat, None),
             out_mask=layer.self_attn.inner_forward(bbox, feat, mask), mask=mask, **common)
        qf, cls, box = layer(bbox, feat, feats_cf, None, metas_run)
        save('g6_layer_T2', out_feat=qf, out_cls=cls, out_bbox=box, **common)

        model8, params8 = build(8, 4, seed=4)
        x_mix = torch.randn(1, 20, 4, 32, 64, generator=torch.Generator().manual_seed(43))
        q8 = torch.randn(1, 20, 256, generator=torch.Generator().manual_seed(44))
        save('g4_mixing_T8', x=x_mix, query_feat=q8, out=model8.decoder.decoder_layer.mixing.inner_forward(x_mix, q8),
             seeds=np.array([4]), params_checksum=S.checksum(params8))

        # ---- G7: full 6-layer decoder through SparseBEVTransformer.forward --------------------------------
        for tag, T, Q, pyr, L, B in (('c1', 1, 100, 'r50_704x256', 4, 1), ('c2small', 8, 100, 'tiny', 4, 1), ('L5', 2, 36, 'tiny5', 5, 2)):
            model, params = build(T, L, seed=7)
            metas, ih, iw, sizes = metas_for(B, T, pyr)
            bbox, feat = S.make_queries(B, Q, seed=71)
            feats = S.make_features(B, T, sizes, seed=72)
            per_layer = []          # the reference layer's query_feat output at every stage (teacher forcing in tests)
            hook = model.decoder.decoder_layer.register_forward_hook(lambda mod, inp, out: per_layer.append(out[0].clone()))
            cls, box = model(bbox, feat, [f.clone() for f in feats], None, copy.deepcopy(metas))
            hook.remove()
            save('g7_decoder_' + tag, query_bbox=bbox, query_feat=feat, out_cls=cls, out_bbox=box,
                 out_feat=torch.stack(per_layer),
                 cfg=np.array([B, Q, T, L]), pyramid=np.array(pyr), seeds=np.array([7, 71, 72]),
                 timestamps=np.array([m['img_timestamp'] for m in metas]),
                 params_checksum=S.checksum(params), feats_checksum=S.checksum(feats))


def main_head():
    """G9: the reference's NMSFreeCoder.decode (models/bbox/coders/nms_free_coder.py) and denormalize_bbox
    (models/bbox/utils.py), executed as they are -- the only stand-ins are mmdet's registry decorator and the empty
    BaseBBoxCoder base class the coder file imports.  `python tests/golden/make_golden.py head` writes only G9."""
    class _Registry:
        def register_module(self):
            return lambda cls: cls

    _stub('models', REF + '/models'); _stub('models.bbox', REF + '/models/bbox'); _stub('models.bbox.coders', REF + '/models/bbox/coders')
    _stub('mmdet'); _stub('mmdet.core'); _stub('mmdet.core.bbox', BaseBBoxCoder=object)
    _stub('mmdet.core.bbox.builder', BBOX_CODERS=_Registry())
    coder_mod = importlib.import_module('models.bbox.coders.nms_free_coder')
    butils = importlib.import_module('models.bbox.utils')
    pc_range = S.PC_RANGE
    post = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]                     # configs/r50_nuimg_704x256.py:80-87
    for tag, B, Q, NC, max_num, thr in (('c2', 2, 900, 10, 300, 0.05), ('small', 3, 36, 10, 100, None), ('few', 1, 16, 3, 48, 0.3)):
        g = torch.Generator().manual_seed(900 + Q)
        cls = torch.randn(2, B, Q, NC, generator=g) * 2 - 2.5           # mostly below the threshold, like a trained head
        box = torch.randn(2, B, Q, 10, generator=g)
        box[..., 0:2] = box[..., 0:2] * 40                              # some centres outside +-61.2 m
        box[..., 4] = box[..., 4] * 6                                   # some cz outside +-10 m
        box[..., 2:4] = box[..., 2:4] * 0.3 + 0.5
        box[..., 5] = box[..., 5] * 0.3 + 0.4
        coder = coder_mod.NMSFreeCoder(pc_range, post_center_range=post, max_num=max_num, score_threshold=thr, num_classes=NC)
        dec = coder.decode({'all_cls_scores': cls, 'all_bbox_preds': box})
        arrays = dict(cls=cls, box=box, cfg=np.array([B, Q, NC, max_num]), thr=np.array(-1.0 if thr is None else thr),
                      post=np.array(post), denorm_all=butils.denormalize_bbox(box[-1]))
        for i, d in enumerate(dec):
            arrays['bboxes%d' % i], arrays['scores%d' % i], arrays['labels%d' % i] = d['bboxes'], d['scores'], d['labels']
        save('g9_nms_free_' + tag, **arrays)


def main_version():
    """G10: the reference's make_sample_points / rotation_3d_in_axis under VERSION.name = 'v0.17.1' (models/utils.py:66-77),
    the convention old checkpoints select (val.py:128-129).  `python tests/golden/make_golden.py version` writes only G10."""
    tr, smp, wrap, utils = import_reference()
    g = torch.Generator().manual_seed(1017)
    bbox, _ = S.make_queries(2, 16, seed=1018)
    bbox[..., 6:8] = torch.randn(2, 16, 2, generator=g)
    offset = torch.randn(2, 16, 16, 3, generator=g)
    out = {}
    for name in ('v1.0.0', 'v0.17.1'):
        utils.VERSION.name = name
        out[name] = smp.make_sample_points(bbox, offset, S.PC_RANGE)
    utils.VERSION.name = 'v1.0.0'
    assert (out['v1.0.0'] - out['v0.17.1']).abs().max() > 1e-2
    save('g10_sample_points_versions', query_bbox=bbox, offset=offset, pts_v1=out['v1.0.0'], pts_v017=out['v0.17.1'])



def plant_nonfinite_borders(feats_cl, g):
    """Inf / NaN in border pixels (what an fp16 backbone can leave, val.py:115): per (sample-batch entry, level) the whole outer
    ring, or single border pixels in some channels, or nothing -- the SAME pixels in all 6 views of the entry, because the native
    path's grid_sample is trilinear over the view axis and reads the neighbouring view with weight 0 (0 x Inf = NaN) where the CUDA
    kernel rounds to one view: with identical patterns both read a bad pixel for exactly the same points.  Interior pixels stay
    finite."""
    for f in feats_cl:
        Bp, N, H, W, C = f.shape
        if min(H, W) < 4:                   # (a 2 x 6 map is border only)
            continue
        for b in range(Bp):
            r = b % 3
            bad = float('inf') if b % 2 == 0 else float('nan')
            if r == 0:                      # the whole ring
                f[b, :, 0], f[b, :, H - 1] = bad, bad
                f[b, :, :, 0], f[b, :, :, W - 1] = bad, bad
            elif r == 1:                    # one corner pixel and one edge pixel, some channels only
                f[b, :, 0, 0, ::2] = bad
                f[b, :, H - 1, W // 2, 1::3] = bad
    return feats_cl


def main_nonfinite():
    """G12: the reference's native-PyTorch sampler (csrc/wrapper.py:14-38 -> F.grid_sample, zeros padding) on maps whose BORDER
    pixels hold Inf / NaN, sample points from far outside to well inside: an out-of-map bilinear corner is never read
    (msmv_sampling_forward.cu:47-66; grid_sample's within-bounds test), so a point wholly outside a map gets exactly 0 from it
    and only points whose footprint really covers a bad pixel turn non-finite.  WHICH elements are non-finite is what this fixture
    pins (the kind may differ: the native path adds 0 x Inf = NaN from the neighbouring view, see plant_nonfinite_borders).  `python tests/golden/make_golden.py nonfinite`."""
    tr, smp, wrap, utils = import_reference()
    assert wrap.MSMV_CUDA is False
    for tag, pyr, C, Bp in (('L4_C64', 'tiny', 64, 3), ('L5_C64', 'tiny5', 64, 2)):
        g = torch.Generator().manual_seed(1200 + len(S.PYRAMIDS[pyr][2]))
        sizes = S.PYRAMIDS[pyr][2]
        Q, P, L = 24, 4, len(sizes)
        feats_cl = plant_nonfinite_borders([torch.randn(Bp, 6, h, w, C, generator=g) for h, w in sizes], g)
        # random coordinates only: on an exact pixel / border coordinate (edge_locs' specials) the native path's grid arithmetic and the
        # CUDA kernel's round differently by an ulp and legitimately pick different corners
        loc = torch.rand(Bp, Q, P, 3, generator=g)
        loc[..., 2] = torch.randint(0, 6, (Bp, Q, P), generator=g).float() / 5
        loc[:, Q // 2:, :, :2] = loc[:, Q // 2:, :, :2] * 1.6 - 0.3          # half the queries: x, y in [-0.3, 1.3]
        loc[:, :2, :, 0] = -3.0                                              # far outside
        loc[:, 2:4, :, 1] = 4.0
        wts = torch.softmax(torch.randn(Bp, Q, P, L, generator=g), -1)
        out = wrap.msmv_sampling_pytorch(cf_from_cl(feats_cl), loc, wts)
        nf = ~torch.isfinite(out)
        print('  G12 %s: %.1f %% of the outputs non-finite' % (tag, 100 * nf.float().mean()))
        assert 0.02 < nf.float().mean() < 0.9
        save('g12_msmv_nonfinite_' + tag, loc=loc, weights=wts, out=out, sizes=np.array(sizes),
             **{'feat%d' % i: f for i, f in enumerate(feats_cl)})


sample_indices = S.grad_sample_indices


def main_train():
    """G11: gradients of the REFERENCE decoder (its own modules, autograd, activation checkpoints -- models/
    sparsebev_transformer.py:231-234,313-317,383-387) in train() mode with the mmcv dropouts set to 0 (RNG streams cannot
    be matched), for random cotangents on (cls_scores, bbox_preds): d/d query_feat, query_bbox, all 48 parameters and the
    feature maps.  Big gradients are stored as a seeded subsample + norm + sum.  `python tests/golden/make_golden.py train`."""
    import copy
    tr, smp, wrap, utils = import_reference()
    assert wrap.MSMV_CUDA is False
    for tag, n_layers, B, Q, T, L, pyr in (('L2', 2, 2, 36, 2, 4, 'tiny'), ('L6', 6, 1, 36, 2, 4, 'tiny')):
        params = S.make_params(11, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
        m = tr.SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=n_layers, num_levels=L, num_classes=10,
                                    code_size=10, pc_range=S.PC_RANGE)
        m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
        m.train()
        layer = m.decoder.decoder_layer
        layer.self_attn.attention.attn.dropout = 0.0
        for mod in layer.ffn.modules():
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
        ih, iw, sizes = S.PYRAMIDS[pyr]
        metas = S.make_img_metas(B, T, ih, iw)
        for b, mm in enumerate(metas):
            mm['img_timestamp'] = [ts + 0.003 * ((i * 7 + b) % 6) for i, ts in enumerate(mm['img_timestamp'])]
        bbox, feat = S.make_queries(B, Q, seed=111)
        feats = S.make_features(B, T, sizes, seed=112)
        g = torch.Generator().manual_seed(113)
        cot_cls = torch.randn(n_layers, B, Q, 10, generator=g)
        cot_box = torch.randn(n_layers, B, Q, 10, generator=g)
        bbox_g, feat_g = bbox.clone().requires_grad_(True), feat.clone().requires_grad_(True)
        feats_g = [f.clone().requires_grad_(True) for f in feats]
        per_layer = []          # the layer's query_feat output at every stage (value forcing in the tests, see test_gpu_backward.py)
        hook = layer.register_forward_hook(lambda mod, inp, out: per_layer.append(out[0].detach().clone()))
        with torch.enable_grad():
            cls, box = m(bbox_g, feat_g, list(feats_g), None, copy.deepcopy(metas))
            loss = (cls * cot_cls).sum() + (box * cot_box).sum()
            loss.backward()
        hook.remove()
        arrays = dict(query_bbox=bbox, query_feat=feat, cot_cls=cot_cls, cot_box=cot_box, out_cls=cls, out_bbox=box,
                      out_feat=torch.stack(per_layer[:n_layers]),
                      grad_query_bbox=bbox_g.grad, grad_query_feat=feat_g.grad,
                      cfg=np.array([B, Q, T, L, n_layers]), pyramid=np.array(pyr), seeds=np.array([11, 111, 112, 113]),
                      timestamps=np.array([mm['img_timestamp'] for mm in metas]),
                      params_checksum=S.checksum(params), feats_checksum=S.checksum(feats))
        named = [(k[len(PREFIX):], p.grad) for k, p in m.named_parameters()] + [('feat%d' % i, f.grad) for i, f in enumerate(feats_g)]
        for name, gr in named:
            assert gr is not None, name
            idx = sample_indices(gr.numel())
            key = 'g.' + name
            arrays[key + '.norm'] = gr.double().norm().float()
            arrays[key + '.sum'] = gr.double().sum().float()
            arrays[key] = gr if idx is None else gr.reshape(-1)[idx]
        print('  G11 %s: loss %.4f, |g query_feat| %.3e, |g pg_w| %.3e, |g feat0| %.3e' % (
            tag, float(loss), float(feat_g.grad.norm()), float(dict(named)['mixing.parameter_generator.weight'].norm()), float(feats_g[0].grad.norm())))
        save('g11_train_' + tag, **arrays)


def main_yardstick():
    """G13: how far does the REFERENCE itself drift from itself over 6 free-running layers?  The decoder (random-init weights, no
    trained contraction) amplifies fp32 rounding noise layer by layer, so a free-running comparison of ANY two fp32 implementations
    diverges; the question a parity test has to answer is whether ours diverges from the reference more than the reference's own two
    samplers diverge from each other on the same inputs.  Recorded here, at G7's c1 and c2small inputs (same seeds; the native run is
    asserted equal to the committed G7 fixture; also at G7's 5-level L5 case):
      * `native`  -- the decoder on `msmv_sampling_pytorch` (models/csrc/wrapper.py:14-38; MSMV_CUDA False: channel-first regroup,
                     F.grid_sample, trilinear over the view axis) = G7's outputs;
      * `kernel`  -- the SAME reference decoder with `MSMV_CUDA = True` (models/sparsebev_transformer.py:78-83: channel-last regroup) and
                     `msmv_sampling` (models/csrc/wrapper.py:87-93) answered by the restatement of the CUDA kernel's semantics
                     (oracle.msmv_sampling_kernel_semantics: msmv_sampling_forward.cu:27-164 -- round-to-one-view, per-corner zero
                     padding; the CUDA extension itself cannot be built here) -- what a reference user with the compiled extension runs;
      * `ulp`     -- the native run again with query_feat nudged by ONE fp32 ulp in every other element (four nudges: offset 0 / 1,
                     up / down; their envelope): the conditioning of the 6-layer map itself, independent of any sampler.
    Stored: the kernel run's per-layer cls / bbox / query_feat and the per-layer max-abs divergences native-vs-kernel, native-vs-ulp.
    `python tests/golden/make_golden.py yardstick` writes only G13."""
    import copy
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tr, smp, wrap, utils = import_reference()
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import sparsebev_oracle as O

    def build(T, L, seed):
        params = S.make_params(seed, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
        m = tr.SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=6, num_levels=L, num_classes=10, code_size=10, pc_range=S.PC_RANGE)
        m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
        return m.eval(), params

    def run(model, bbox, feat, feats, metas):
        per_layer = []
        hook = model.decoder.decoder_layer.register_forward_hook(lambda mod, inp, out: per_layer.append(out[0].clone()))
        with torch.no_grad():
            cls, box = model(bbox, feat, [f.clone() for f in feats], None, copy.deepcopy(metas))
        hook.remove()
        return cls, box, torch.stack(per_layer)

    for tag, T, Q, pyr, L, B in (('c1', 1, 100, 'r50_704x256', 4, 1), ('c2small', 8, 100, 'tiny', 4, 1), ('L5', 2, 36, 'tiny5', 5, 2)):
        model, params = build(T, L, seed=7)
        ih, iw, sizes = S.PYRAMIDS[pyr]
        metas = S.make_img_metas(B, T, ih, iw)
        for b, m in enumerate(metas):
            m['img_timestamp'] = [ts + 0.003 * ((i * 7 + b) % 6) for i, ts in enumerate(m['img_timestamp'])]
        bbox, feat = S.make_queries(B, Q, seed=71)
        feats = S.make_features(B, T, sizes, seed=72)
        assert wrap.MSMV_CUDA is False and tr.MSMV_CUDA is False
        nat = run(model, bbox, feat, feats, metas)
        g7 = np.load(os.path.join(HERE, 'g7_decoder_%s.npz' % tag))
        assert np.array_equal(nat[0].numpy(), g7['out_cls']) and np.array_equal(nat[1].numpy(), g7['out_bbox']), 'native run != committed G7'
        # the reference with its CUDA-path switches thrown, the extension's entry point answered by the kernel-semantics restatement
        saved = (tr.MSMV_CUDA, smp.msmv_sampling)
        tr.MSMV_CUDA = True
        smp.msmv_sampling = lambda mlvl, loc, w: O.msmv_sampling_kernel_semantics(list(mlvl), loc, w)
        try:
            ker = run(model, bbox, feat, feats, metas)
        finally:
            tr.MSMV_CUDA, smp.msmv_sampling = saved
        # four one-ulp nudges of query_feat (every other element from offset 0 / 1, towards +Inf / -Inf): a chaotic amplification has
        # a wide spread between realisations, the envelope of several is the yardstick
        ulps = []
        for off, way in ((0, 1.0), (1, 1.0), (0, -1.0), (1, -1.0)):
            nudged = feat.clone()
            flat = nudged.view(-1)
            flat[off::2] = torch.nextafter(flat[off::2], torch.full_like(flat[off::2], way * float('inf')))
            ulps.append(run(model, bbox, nudged, feats, metas))

        def div(a, b):
            return np.array([(a[i] - b[i]).abs().max().item() for i in range(a.shape[0])], dtype=np.float64)
        d = {}
        for k, what in enumerate(('cls', 'bbox', 'feat')):
            d['div_%s_kernel' % what] = div(nat[k], ker[k])
            d['div_%s_ulp' % what] = np.max(np.stack([div(nat[k], u[k]) for u in ulps]), axis=0)
        for k in sorted(d):
            print('  G13 %-8s %-16s %s' % (tag, k, ' '.join('%.2e' % v for v in d[k])))
        save('g13_yardstick_' + tag, cfg=np.array([B, Q, T, L]), pyramid=np.array(pyr), seeds=np.array([7, 71, 72]),
             timestamps=np.array([m['img_timestamp'] for m in metas]),
             kernel_cls=ker[0], kernel_bbox=ker[1], kernel_feat=ker[2],
             params_checksum=S.checksum(params), feats_checksum=S.checksum(feats), **d)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'head':
        main_head()
    elif len(sys.argv) > 1 and sys.argv[1] == 'version':
        main_version()
    elif len(sys.argv) > 1 and sys.argv[1] == 'train':
        main_train()
    elif len(sys.argv) > 1 and sys.argv[1] == 'nonfinite':
        main_nonfinite()
    elif len(sys.argv) > 1 and sys.argv[1] == 'yardstick':
        main_yardstick()
    else:
        main()
        main_head()
        main_version()
        main_train()
        main_nonfinite()
        main_yardstick()
