"""Drop-in ``SparseBEVTransformer`` for ``SparseBEVHead`` -- MI355X-native decoder hot path.

Interface parity with the reference:
  * same registry name and constructor kwargs as ``models/sparsebev_transformer.py:16-26``
    (``dict(type='SparseBEVTransformer', embed_dims, num_frames, num_points, num_layers, num_levels,
    num_classes, code_size, pc_range)``; ``init_cfg`` must be None);
  * same ``forward(query_bbox, query_feat, mlvl_feats, attn_mask, img_metas) -> (cls_scores, bbox_preds)``
    (``:32-38``), same ``init_weights()`` (``:28-30,146-153,206-208,265-268,348-349``);
  * identical ``state_dict`` keys (48 tensors under ``decoder.decoder_layer.``), so reference checkpoints
    load with ``strict=True``.
The module tree below exists only to own parameters under those names; the arithmetic is done by the HIP
kernels in libsbev_hip.so through ``sparsebev_amd.ops`` / ``sparsebev_amd.dense``.  Differences on purpose:
inputs are never mutated (the reference overwrites ``mlvl_feats[lvl]`` and ``img_metas[0]``, ``:65,70,85``),
and the 2x-feature-bytes regroup copy (``:73-85``) is replaced by one NCHW->NHWC relayout (or nothing at
all when the neck already produces channels-last features).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import autograd as AG
from . import dense, ops
from .utils import DUMP, VERSION

try:  # optional: register with mmdet's registry when the OpenMMLab stack is present
    from mmcv.runner import BaseModule as _Base
    from mmdet.models.utils.builder import TRANSFORMER as _REGISTRY
except Exception:  # noqa: BLE001  (mmcv / mmdet are absent in the build image)
    _REGISTRY = None

    class _Base(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()

N_VIEWS, N_GROUPS, N_HEADS, OUT_POINTS, FFN_CHANNELS = 6, 4, 8, 128, 512


def _register(cls):
    if _REGISTRY is not None:
        try:
            return _REGISTRY.register_module()(cls)
        except KeyError:        # the reference's own class is already registered under this name
            return cls
    return cls


class _AttentionParams(nn.Module):
    """Parameter holder with mmcv-1.6.0 ``MultiheadAttention`` key names: ``attn.in_proj_weight`` etc."""

    def __init__(self, embed_dims, num_heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, dropout=0.0)


class _FFNParams(nn.Module):
    """mmcv-1.6.0 ``FFN`` key names: ``layers.0.0`` (Linear D->F) and ``layers.1`` (Linear F->D)."""

    def __init__(self, embed_dims, feedforward_channels):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(0.1)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(0.1))


class SparseBEVSelfAttention(_Base):
    """Scale-adaptive self attention, models/sparsebev_transformer.py:196-248."""

    def __init__(self, embed_dims=256, num_heads=8, dropout=0.1, pc_range=(), init_cfg=None):
        super().__init__(init_cfg)
        self.pc_range = list(pc_range)
        self.num_heads = num_heads
        self.attn_drop = float(dropout)      # mmcv MultiheadAttention(embed_dims, num_heads, attn_drop=dropout): active in train() only
        self.attention = _AttentionParams(embed_dims, num_heads)
        self.gen_tau = nn.Linear(embed_dims, num_heads)

    @torch.no_grad()
    def init_weights(self):
        nn.init.zeros_(self.gen_tau.weight)
        nn.init.uniform_(self.gen_tau.bias, 0.0, 2.0)

    def in_proj_packed(self):
        """(weight, bias) of the one in-projection GEMM: q | k | v | tau rows."""
        a = self.attention.attn
        return dense._cat_rows(a.in_proj_weight, self.gen_tau.weight), dense._cat_rows(a.in_proj_bias, self.gen_tau.bias)

    def forward(self, query_bbox, query_feat, pre_attn_mask=None, ln=None, qkvt=None):
        a = self.attention.attn
        if DUMP.enabled:     # sasa_tau tap (models/sparsebev_transformer.py:218-219); debug path only
            w4 = torch.cat([self.gen_tau.weight, self.gen_tau.weight.new_zeros((-self.num_heads) % 4, query_feat.shape[-1])], 0)
            b4 = torch.cat([self.gen_tau.bias, self.gen_tau.bias.new_zeros((-self.num_heads) % 4)], 0)
            DUMP.save('sasa_tau', dense.linear(query_feat, w4, b4)[..., :self.num_heads])
        return dense.scale_adaptive_self_attention(
            query_bbox, query_feat, self.pc_range, self.num_heads,
            a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
            self.gen_tau.weight, self.gen_tau.bias, pre_attn_mask, ln=ln, qkvt=qkvt)


class SparseBEVSampling(_Base):
    """Adaptive spatio-temporal sampling, models/sparsebev_transformer.py:251-317."""

    def __init__(self, embed_dims=256, num_frames=4, num_groups=4, num_points=8, num_levels=4, pc_range=(), init_cfg=None):
        super().__init__(init_cfg)
        self.num_frames, self.num_points, self.num_groups, self.num_levels = num_frames, num_points, num_groups, num_levels
        self.pc_range = list(pc_range)
        self.sampling_offset = nn.Linear(embed_dims, num_groups * num_points * 3)
        self.scale_weights = nn.Linear(embed_dims, num_groups * num_points * num_levels)

    @torch.no_grad()
    def init_weights(self):
        nn.init.zeros_(self.sampling_offset.weight)
        nn.init.uniform_(self.sampling_offset.bias, -0.5, 0.5)

    def packed(self):
        """(weight, bias) of the one Linear for both generators: [256] -> G*P*3 offsets | G*P*L level logits."""
        return (dense._cat_rows(self.sampling_offset.weight, self.scale_weights.weight),
                dense._cat_rows(self.sampling_offset.bias, self.scale_weights.bias))

    def forward(self, query_bbox, query_feat, feats, ctx, both=None):
        """feats: FeaturePyramid (channels-last, resident); ctx: DecoderContext.  -> [B,Q,G,T*P,C]"""
        T, G, P, L = self.num_frames, self.num_groups, self.num_points, self.num_levels
        if both is None:                                            # else: produced by ln_linear() together with query_feat
            both = dense.linear(query_feat, *self.packed())
        n_off = G * P * 3
        offset, logits = both[..., :n_off], both[..., n_off:]       # column slices of the packed rows (no copy)
        pts, w_bp = ops.sampling_front(query_bbox, offset, logits, ctx.time_diff, self.pc_range, T, G, P, L)
        if DUMP.enabled:
            loc, uvh, valid, _ = ops.project_select(pts, ctx.lidar2img, ctx.image_h, ctx.image_w, G, P, dump=True)
            DUMP.save('sample_points_cam', uvh)
            DUMP.save('sample_points_cam_valid_mask', valid.float())
        else:
            loc = ops.project_select(pts, ctx.lidar2img, ctx.image_h, ctx.image_w, G, P)
        return feats.sample(loc, w_bp, T, G)


class AdaptiveMixing(nn.Module):
    """Adaptive channel + point mixing, models/sparsebev_transformer.py:320-387."""

    def __init__(self, in_dim, in_points, n_groups=1, query_dim=None, out_dim=None, out_points=None):
        super().__init__()
        out_dim = out_dim if out_dim is not None else in_dim
        out_points = out_points if out_points is not None else in_points
        query_dim = query_dim if query_dim is not None else in_dim
        self.query_dim, self.in_dim, self.in_points, self.n_groups = query_dim, in_dim, in_points, n_groups
        self.out_dim, self.out_points = out_dim, out_points
        self.eff_in_dim, self.eff_out_dim = in_dim // n_groups, out_dim // n_groups
        self.m_parameters = self.eff_in_dim * self.eff_out_dim
        self.s_parameters = self.in_points * self.out_points
        self.total_parameters = self.m_parameters + self.s_parameters
        self.parameter_generator = nn.Linear(self.query_dim, self.n_groups * self.total_parameters)
        self.out_proj = nn.Linear(self.eff_out_dim * self.out_points * self.n_groups, self.query_dim)

    @torch.no_grad()
    def init_weights(self):
        nn.init.zeros_(self.parameter_generator.weight)

    def forward(self, x, query, ln=None):
        return dense.adaptive_mixing(x, query, self.parameter_generator.weight, self.parameter_generator.bias,
                                     self.out_proj.weight, self.out_proj.bias, self.out_points, ln=ln)


class SparseBEVTransformerDecoderLayer(_Base):
    """models/sparsebev_transformer.py:104-193."""

    def __init__(self, embed_dims, num_frames=8, num_points=4, num_levels=4, num_classes=10, code_size=10,
                 num_cls_fcs=2, num_reg_fcs=2, pc_range=(), init_cfg=None):
        super().__init__(init_cfg)
        if code_size != 10:     # sasa / sampling_front / refine / linear3 kernels read query_bbox rows of 10 floats (every reference config uses 10)
            raise ValueError('sparsebev_amd is built for code_size == 10 (cx, cy, cz, w, l, h, sin, cos, vx, vy); got %d' % code_size)
        self.embed_dims, self.num_classes, self.code_size, self.pc_range = embed_dims, num_classes, code_size, list(pc_range)
        D = embed_dims
        self.position_encoder = nn.Sequential(nn.Linear(3, D), nn.LayerNorm(D), nn.ReLU(inplace=True),
                                              nn.Linear(D, D), nn.LayerNorm(D), nn.ReLU(inplace=True))
        self.self_attn = SparseBEVSelfAttention(D, num_heads=N_HEADS, dropout=0.1, pc_range=pc_range)
        self.sampling = SparseBEVSampling(D, num_frames=num_frames, num_groups=N_GROUPS, num_points=num_points,
                                          num_levels=num_levels, pc_range=pc_range)
        self.mixing = AdaptiveMixing(in_dim=D, in_points=num_points * num_frames, n_groups=N_GROUPS, out_points=OUT_POINTS)
        self.ffn = _FFNParams(D, FFN_CHANNELS)
        self.ffn_drop = 0.1                  # mmcv FFN(ffn_drop=0.1): active in train() only
        self.recompute_mixing = False        # training: keep the dynamic parameters / mixed activations (236 MB per layer and sample at
                                             # T = 8) instead of re-running generator GEMM + mixing in backward like the reference's
                                             # checkpoint (:383-387); True trades 183 us per layer for that memory
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(D), nn.LayerNorm(D), nn.LayerNorm(D)
        cls = []
        for _ in range(num_cls_fcs):
            cls += [nn.Linear(D, D), nn.LayerNorm(D), nn.ReLU(inplace=True)]
        cls.append(nn.Linear(D, num_classes))
        self.cls_branch = nn.Sequential(*cls)
        reg = []
        for _ in range(num_reg_fcs):
            reg += [nn.Linear(D, D), nn.ReLU(inplace=True)]
        reg.append(nn.Linear(D, code_size))
        self.reg_branch = nn.Sequential(*reg)

    @torch.no_grad()
    def init_weights(self):
        self.self_attn.init_weights()
        self.sampling.init_weights()
        self.mixing.init_weights()
        nn.init.constant_(self.cls_branch[-1].bias, float(-math.log((1 - 0.01) / 0.01)))   # bias_init_with_prob(0.01)

    def packed_train_weights(self):
        """The two concatenated Linears of the differentiable path (attention in-projection | gen_tau, sampling_offset | scale_weights),
        built ONCE per decoder call and shared by its layers (the 6 layers share the parameters): 4 `cat` launches and 2 W^T transposes
        per step instead of 24 and 12; gradients reach the parameters through the one cat node each."""
        sa, smp = self.self_attn, self.sampling
        att = sa.attention.attn
        D, H = self.embed_dims, sa.num_heads
        pad = (-(3 * D + H)) % 4
        in_w = torch.cat([att.in_proj_weight, sa.gen_tau.weight] + ([att.in_proj_weight.new_zeros(pad, D)] if pad else []), 0)
        in_b = torch.cat([att.in_proj_bias, sa.gen_tau.bias] + ([att.in_proj_bias.new_zeros(pad)] if pad else []), 0)
        samp_w = torch.cat([smp.sampling_offset.weight, smp.scale_weights.weight], 0)
        samp_b = torch.cat([smp.sampling_offset.bias, smp.scale_weights.bias], 0)
        return in_w, in_b, samp_w, samp_b

    def forward_train(self, query_bbox, query_feat, feats, attn_mask, ctx, feat_token=None, packed=None, tap=None):
        """The same layer with every op as a differentiable node (sparsebev_amd.autograd: HIP forward + HIP backward),
        unfused where a fused inference launch would hide an activation the backward needs.  Dropout (attention
        probabilities 0.1, the two FFN dropouts 0.1 -- mmcv defaults the reference's layer is built with,
        models/sparsebev_transformer.py:125,202) is active like in the reference's train() mode; set ``self.self_attn.attn_drop``
        / ``self.ffn_drop`` to 0 for deterministic gradients."""
        pe, sa, smp, mix = self.position_encoder, self.self_attn, self.sampling, self.mixing
        att = sa.attention.attn
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (sa.attn_drop > 0 or self.ffn_drop > 0) else 0
        tk = (tap, tap.token) if (tap is not None and tap.token is not None) else ()
        pos = AG.Linear3LnRelu.apply(query_bbox, pe[0].weight, pe[0].bias, pe[1].weight, pe[1].bias, *tk)
        x = AG.layer_norm(AG.linear(pos, pe[3].weight, pe[3].bias, tap=tap), pe[4].weight, pe[4].bias, relu=True, add_after=query_feat, tap=tap)
        # self attention (+ identity), norm1
        D, H = self.embed_dims, sa.num_heads
        in_w, in_b, samp_w, samp_b = packed if packed is not None else self.packed_train_weights()
        qkvt = AG.linear(x, in_w, in_b, tap=tap)
        mask = attn_mask.to(device=x.device, dtype=torch.uint8).contiguous() if attn_mask is not None else None
        a = AG.SasaCore.apply(qkvt, query_bbox, mask, tuple(sa.pc_range), H, sa.attn_drop, seed)
        x = AG.layer_norm(AG.linear(a, att.out_proj.weight, att.out_proj.bias, residual=x, tap=tap), self.norm1.weight, self.norm1.bias, tap=tap)
        # adaptive spatio-temporal sampling
        both = AG.linear(x, samp_w, samp_b, tap=tap)
        cfg = (smp.num_frames, smp.num_groups, smp.num_points, smp.num_levels, tuple(smp.pc_range))
        sampled = AG.Sampling.apply(query_bbox, both, feats, ctx, cfg, feat_token)     # feat_token: AG.feature_token (None: frozen features)
        # adaptive mixing (+ identity), norm2
        x = AG.layer_norm(AG.AdaptiveMixing.apply(sampled, x, mix.parameter_generator.weight, mix.parameter_generator.bias,
                                                  mix.out_proj.weight, mix.out_proj.bias, mix.out_points, self.recompute_mixing,
                                                  getattr(self, 'train_gemm_f16', False), *tk),
                          self.norm2.weight, self.norm2.bias, tap=tap)
        # FFN (+ identity), norm3
        f0, f1 = self.ffn.layers[0][0], self.ffn.layers[1]
        h = AG.dropout(AG.linear(x, f0.weight, f0.bias, relu=True, tap=tap), self.ffn_drop, seed + 1)
        if self.ffn_drop > 0:
            # the torch add below is the `identity +` of mmcv's FFN around a dropped-out branch (autograd plumbing)
            x = AG.layer_norm(x + AG.dropout(AG.linear(h, f1.weight, f1.bias, tap=tap), self.ffn_drop, seed + 2), self.norm3.weight, self.norm3.bias, tap=tap)
        else:
            x = AG.layer_norm(AG.linear(h, f1.weight, f1.bias, residual=x, tap=tap), self.norm3.weight, self.norm3.bias, tap=tap)
        cb, rb = self.cls_branch, self.reg_branch
        if os.environ.get('SBEV_NO_TRAIN_PAIRS') == '1':       # A/B: one launch per Linear (rounds 2-5)
            c = AG.layer_norm(AG.linear(x, cb[0].weight, cb[0].bias, tap=tap), cb[1].weight, cb[1].bias, relu=True, tap=tap)
            c = AG.layer_norm(AG.linear(c, cb[3].weight, cb[3].bias, tap=tap), cb[4].weight, cb[4].bias, relu=True, tap=tap)
            cls_score = AG.linear(c, cb[6].weight, cb[6].bias, tap=tap)
            r = AG.linear(x, rb[0].weight, rb[0].bias, relu=True, tap=tap)
            r = AG.linear(r, rb[2].weight, rb[2].bias, relu=True, tap=tap)
            reg = AG.linear(r, rb[4].weight, rb[4].bias, tap=tap)
        else:
            # the two branch heads level by level, each level ONE grouped launch forward and one for the two grad_x products backward
            # (AG.LinearPair; the inference runtime groups the same way: csrc/decoder.hip `grouped`)
            c, r = AG.linear_pair(x, cb[0].weight, cb[0].bias, False, x, rb[0].weight, rb[0].bias, True, tap=tap)
            c = AG.layer_norm(c, cb[1].weight, cb[1].bias, relu=True, tap=tap)
            c, r = AG.linear_pair(c, cb[3].weight, cb[3].bias, False, r, rb[2].weight, rb[2].bias, True, tap=tap)
            c = AG.layer_norm(c, cb[4].weight, cb[4].bias, relu=True, tap=tap)
            cls_score, reg = AG.linear_pair(c, cb[6].weight, cb[6].bias, False, r, rb[4].weight, rb[4].bias, False, tap=tap)
        bbox_pred = AG.RefineBbox.apply(query_bbox, reg, ctx.vel_div)
        return x, cls_score, bbox_pred

    def forward(self, query_bbox, query_feat, feats, attn_mask, ctx):
        pe = self.position_encoder
        pos = dense.linear_ln_relu(query_bbox, pe[0].weight, pe[0].bias, pe[1].weight, pe[1].bias)   # reads columns 0:3
        # three of the layer's LayerNorms run as the PROLOGUE of the Linear that consumes them (dense.ln_linear, the same
        # launches as the C++ runtime): position-encoder norm (+ ReLU, + query_feat) -> attention in_proj; norm1 ->
        # sampling generators; norm3 -> first Linear of the classification branch
        x, qkvt = dense.ln_linear(dense.linear(pos, pe[3].weight, pe[3].bias), pe[4].weight, pe[4].bias,
                                  *self.self_attn.in_proj_packed(), ln_relu=True, add_after=query_feat)
        pre1 = self.self_attn(query_bbox, x, attn_mask, qkvt=qkvt)                  # x + attention, norm1 follows
        x, both = dense.ln_linear(pre1, self.norm1.weight, self.norm1.bias, *self.sampling.packed())
        sampled = self.sampling(query_bbox, x, feats, ctx, both=both)
        x = self.mixing(sampled, x, ln=(self.norm2.weight, self.norm2.bias))       # norm2 fused into the out-proj reducer
        f0, f1 = self.ffn.layers[0][0], self.ffn.layers[1]
        h = dense.linear(x, f0.weight, f0.bias, relu=True)
        cb, rb = self.cls_branch, self.reg_branch
        x, c = dense.ln_linear(dense.linear(h, f1.weight, f1.bias, residual=x), self.norm3.weight, self.norm3.bias,
                               cb[0].weight, cb[0].bias)
        c = dense.layer_norm(c, cb[1].weight, cb[1].bias, relu=True)
        c = dense.linear_ln_relu(c, cb[3].weight, cb[3].bias, cb[4].weight, cb[4].bias)
        cls_score = dense.linear(c, cb[6].weight, cb[6].bias)
        r = dense.linear(x, rb[0].weight, rb[0].bias, relu=True)
        r = dense.linear(r, rb[2].weight, rb[2].bias, relu=True)
        reg = dense.linear(r, rb[4].weight, rb[4].bias)
        bbox_pred = dense.refine_bbox(query_bbox, reg, ctx.vel_div)
        if DUMP.enabled:
            DUMP.save('query_bbox', ops_decode(query_bbox, self.pc_range))
            DUMP.save('bbox_pred', ops_decode(bbox_pred, self.pc_range))
            DUMP.save('cls_score', torch.sigmoid(cls_score))
        return x, cls_score, bbox_pred


def ops_decode(bbox, pc_range):
    """decode_bbox (models/bbox/utils.py:63-77) for the DUMP taps only (debug path)."""
    lo = bbox.new_tensor(pc_range[:3])
    span = bbox.new_tensor([pc_range[3] - pc_range[0], pc_range[4] - pc_range[1], pc_range[5] - pc_range[2]])
    return torch.cat([bbox[..., :3] * span + lo, bbox[..., 3:6].exp(), torch.atan2(bbox[..., 6:7], bbox[..., 7:8]),
                      bbox[..., 8:10]], dim=-1)


class FeaturePyramid:
    """Channels-last, device-resident view of the FPN output for the sampler.

    Built from the reference layout ``list[L] of [B, T*N, G*C, H, W]`` (models/sparsebev.py:126-131).  If a
    level is already channels-last in memory (``permute(0,1,3,4,2)`` contiguous) it is used in place;
    otherwise ONE relayout to ``[B*T*N, H, W, G*C]`` is made.  Group g is addressed as the channel slice
    [g*C, (g+1)*C) inside the kernel, so the reference's per-group regroup copy never exists."""

    def __init__(self, mlvl_feats):
        self.levels = []
        self.copied = 0
        f0 = mlvl_feats[0]
        self.B, TN, self.GC = f0.shape[0], f0.shape[1], f0.shape[2]
        self.T = TN // N_VIEWS
        for f in mlvl_feats:
            if not f.is_cuda:
                raise RuntimeError('SparseBEVTransformer (sparsebev_amd) needs device features; there is no CPU path')
            nhwc = f.permute(0, 1, 3, 4, 2)
            if not nhwc.is_contiguous():
                nhwc = dense.to_channels_last(f)
                self.copied += 1
            self.levels.append(nhwc.reshape(self.B * TN, f.shape[3], f.shape[4], self.GC))

    @classmethod
    def empty_like_nchw(cls, mlvl_feats):
        """Uninitialised channels-last buffers for the on-demand relayout (runtime.DecoderRuntime.forward_lazy): the step writes only
        the units its sample points read."""
        self = cls.__new__(cls)
        f0 = mlvl_feats[0]
        self.B, TN, self.GC = f0.shape[0], f0.shape[1], f0.shape[2]
        self.T = TN // N_VIEWS
        self.copied = 0
        self.levels = [torch.empty(self.B * TN, f.shape[3], f.shape[4], self.GC, device=f.device, dtype=f.dtype) for f in mlvl_feats]
        return self

    def sample(self, loc, w_bp, T, G):
        return ops.msmv_sampling_nhwc(self.levels, self.B, T, G, loc, w_bp, out_layout=ops.OUT_MIX)

    # -- training: gradient wrt the feature maps ---------------------------------------------------------------------------
    # One zero-initialised channels-last buffer per level, shared by the sampler backward of ALL decoder layers of a call
    # (they accumulate into it with atomics) and handed to autograd once, as a permuted view in the caller's [B,TN,C,H,W]
    # axis order: six per-layer 735 MB gradient tensors and their summation never exist.
    def grad_buffers(self):
        if getattr(self, '_grads', None) is None:
            self._grads = [torch.zeros_like(l, dtype=torch.float32) for l in self.levels]
        return self._grads

    def take_feature_grads(self):
        TN = self.T * N_VIEWS
        g = [b.view(self.B, TN, b.shape[1], b.shape[2], self.GC).permute(0, 1, 4, 2, 3) for b in self._grads]
        self._grads = None
        return g

    def sample_backward(self, loc, w_bp, gout, T, G, grad_levels):
        return ops.msmv_sampling_nhwc_backward(self.levels, self.B, T, G, loc, w_bp, gout, grad_levels, grad_layout=ops.OUT_MIX)


class _PinnedUpload:
    """Host -> device upload of the small per-call constants without stalling the host: a pageable `.to(device)` returns
    only once the copy has executed, i.e. after everything queued before it -- the host then cannot run ahead of the
    device and every step starts with an idle gap of one launch latency.  A ring of page-locked staging buffers (one
    event each, waited for only when the ring wraps onto a copy still in flight) keeps the upload asynchronous."""

    DEPTH = 8

    def __init__(self):
        self._rings = {}

    def __call__(self, arr, device, out=None):
        """``out``: an existing device tensor of arr's shape to refresh in place (a captured graph reads it on replay)."""
        t, device = torch.from_numpy(arr), torch.device(device)
        if device.type != 'cuda' or os.environ.get('SBEV_PAGEABLE_UPLOAD') == '1':
            return t.to(device) if out is None else out.copy_(t)
        key = (tuple(arr.shape), arr.dtype.str, device.index)
        ring = self._rings.get(key)
        if ring is None:
            ring = self._rings[key] = {'next': 0, 'slots': [None] * self.DEPTH}
        i = ring['next']
        ring['next'] = (i + 1) % self.DEPTH
        slot = ring['slots'][i]
        if slot is None:
            slot = ring['slots'][i] = [torch.empty(t.shape, dtype=t.dtype).pin_memory(), torch.cuda.Event()]
        else:
            slot[1].synchronize()
        slot[0].copy_(t)
        out = slot[0].to(device, non_blocking=True) if out is None else out.copy_(slot[0], non_blocking=True)
        slot[1].record(torch.cuda.current_stream(device))
        return out


_upload = _PinnedUpload()


class DecoderContext:
    """Per-call constants: time_diff, lidar2img, image size (models/sparsebev_transformer.py:60-70,276)."""

    @staticmethod
    def pack(img_metas, B):
        """Host half: (packed fp32 array, segment offsets / shapes, image_h, image_w) -- time_diff [B,T], lidar2img
        [B,T*N,4,4] and, when T > 1, the velocity divisor [B] in ONE array (each copy is a ~4 us node on the decoder's
        stream), segments padded to 16 bytes."""
        ts = np.array([m['img_timestamp'] for m in img_metas], dtype=np.float64).reshape(B, -1, N_VIEWS)
        td = np.mean(ts[:, :1, :] - ts, axis=-1).astype(np.float32)                # [B,T]; float64 mean then fp32
        l2i = np.asarray([m['lidar2img'] for m in img_metas]).astype(np.float32)   # [B,T*N,4,4]
        image_h, image_w = img_metas[0]['img_shape'][0][:2]
        # velocity divisor of :179-183: time_diff[:,1] with values < 1e-5 replaced by 1 (only when T > 1)
        d = None
        if td.shape[1] > 1:
            d = td[:, 1].copy()
            d[d < 1e-5] = 1.0
        segs = [td.reshape(-1), l2i.reshape(-1)] + ([d] if d is not None else [])
        offs, n = [], 0
        for a in segs:
            offs.append(n)
            n += (a.size + 3) // 4 * 4
        packed = np.zeros(n, dtype=np.float32)
        for a, o in zip(segs, offs):
            packed[o:o + a.size] = a
        layout = (tuple(offs), td.shape, l2i.shape, None if d is None else d.shape)
        return packed, layout, image_h, image_w

    def __init__(self, img_metas, B, device, out=None):
        """``out``: a device buffer of a previous context with the same layout to refresh in place (graph replay)."""
        packed, layout, self.image_h, self.image_w = self.pack(img_metas, B)
        self._bind(packed, layout, device, out)

    @classmethod
    def from_packed(cls, packed, layout, image_h, image_w, device):
        """From an already packed array that may carry extra words behind the constants (runtime.StepGraphs: the pointer table of a
        replayable step travels in the same upload)."""
        self = cls.__new__(cls)
        self.image_h, self.image_w = image_h, image_w
        self._bind(packed, layout, device, None)
        return self

    def _bind(self, packed, layout, device, out):
        self.layout = layout
        self.buffer = dev = _upload(packed, device, out=out)
        offs, td_shape, l2i_shape, d_shape = layout
        n_td, n_l2i = int(np.prod(td_shape)), int(np.prod(l2i_shape))
        self.time_diff = dev[offs[0]:offs[0] + n_td].view(td_shape)
        self.lidar2img = dev[offs[1]:offs[1] + n_l2i].view(l2i_shape)
        self.vel_div = dev[offs[2]:offs[2] + int(np.prod(d_shape))] if d_shape is not None else None


class _Finished(tuple):
    """(cls, bbox) that already went through the runtime's nan_to_num launch (sbev_finish_outputs) and belong to the caller."""


class SparseBEVTransformerDecoder(_Base):
    """models/sparsebev_transformer.py:41-101 (one shared layer applied num_layers times)."""

    def __init__(self, embed_dims, num_frames=8, num_points=4, num_layers=6, num_levels=4, num_classes=10,
                 code_size=10, pc_range=(), init_cfg=None):
        super().__init__(init_cfg)
        self.num_layers, self.pc_range = num_layers, list(pc_range)
        self._runtime = None
        self.overlap = False        # opt-in two-stream fork/join in the C++ runtime (1: generator GEMM || sampling chain, 2: only the
                                    # classification branch aside): measured -3 % samples/s at c2 -- the big kernels fill every CU, and
                                    # the forked path cannot use the grouped branch launches
        self.tap_param_grads = os.environ.get('SBEV_NO_PARAM_TAP') != '1'     # training: parameter gradients collected per call (autograd.Tap)
        self.static_graph = os.environ.get('SBEV_NO_GRAPH') != '1'      # replay a captured hipGraph for repeated identical (pointer-wise) calls
        # the two big mixing GEMMs (runtime.GEMM_MODES).  Default 'f16x3': fp32 operands as scaled fp16 hi + lo images, 3 products,
        # fp32 accumulation on the 16-bit matrix core -- fp32-class: max and rms error against fp64 BELOW the exact f32-input MFMA
        # kernels' at both GEMM shapes, also on inputs spanning 12 binades (tests/test_gpu_bf16s.py), and generator + out-projection
        # within the review's 2 x 60 us at config 2 (DESIGN_HISTORY.md section 9.1).  'f32' = the exact f32-input MFMA kernels (the default of
        # rounds 1-2; SBEV_GEMM_MODE=f32 selects it process-wide); 'bf16x6' = hi + mid + lo bf16 images, 6 products; 'f16x4' = all four
        # fp16 products; 'bf16x3' / 'bf16x3s' = the 2^-16-class modes.  Shapes the split kernels do not cover (embed_dims != 256) run 'f32'.
        self.gemm_mode = os.environ.get('SBEV_GEMM_MODE') or 'f16x3'
        self.value_forcing = None   # tests only: (bbox per layer, feat per layer) recorded from the reference; the differentiable path then
                                    # evaluates layer i+1 AT those values (x + (x_ref - x).detach()) with the autograd graph intact, so that
                                    # multi-layer gradients can be compared at 1e-4 although fp32 rounding noise grows ~5x per layer
        self.decoder_layer = SparseBEVTransformerDecoderLayer(embed_dims, num_frames, num_points, num_levels,
                                                              num_classes, code_size, pc_range=pc_range)

    def _split_gemm_covers(self, rows):
        """whether csrc/gemm_bf16s.hip covers this layer's generator / out-projection shapes (N % 256 == 0, K % 32 == 0, 256 columns out)"""
        from . import _lib
        mix = self.decoder_layer.mixing
        pg, op = mix.parameter_generator.weight, mix.out_proj.weight
        lib = _lib.load()
        return bool(lib.sbev_linear_bf16s_gen_ok(rows, pg.shape[0], pg.shape[1]) and lib.sbev_linear_bf16s_out_ok(rows, op.shape[0], op.shape[1]))

    @torch.no_grad()
    def init_weights(self):
        self.decoder_layer.init_weights()

    def forward(self, query_bbox, query_feat, mlvl_feats, attn_mask, img_metas, layerwise=False, _may_alias=False, _finish=False):
        """Default: the C++ runtime enqueues all layers from one call (csrc/decoder.hip); a caller that passes the SAME
        tensors again (a serving loop refreshing its inputs in place) gets the whole step -- feature relayout + 6 layers -- as
        ONE hipGraph replay from the second identical call on (``static_graph``; runtime.StepGraphs).  ``layerwise=True``
        (and the DUMP debug taps) run the same kernels one Python call at a time -- the path the per-op tests use."""
        B = query_bbox.shape[0]
        inference = not (torch.is_grad_enabled() and (query_bbox.requires_grad or query_feat.requires_grad
                                                      or any(p.requires_grad for p in self.parameters())
                                                      or any(torch.is_tensor(f) and f.requires_grad
                                                             for f in (mlvl_feats if isinstance(mlvl_feats, (list, tuple)) else ()))))
        if not inference and not self.training:
            # eval() under enabled grad (a caller that forgot torch.no_grad()): inputs the autograd path cannot take -- the online
            # frame ring, bf16 feature storage -- run the inference runtime with a one-time warning instead of raising
            lv = mlvl_feats.levels if hasattr(mlvl_feats, 'levels') else mlvl_feats
            if hasattr(mlvl_feats, 'frame_slots') or any(torch.is_tensor(f) and f.dtype != torch.float32 for f in lv):
                if not getattr(self, '_warned_no_grad', False):
                    import warnings
                    warnings.warn('sparsebev_amd: eval-mode call with grad enabled on ring / bf16 features: running the inference '
                                  'runtime (outputs carry no grad_fn); wrap inference in torch.no_grad()')
                    self._warned_no_grad = True
                inference = True
        if inference and not (layerwise or DUMP.enabled):
            from .runtime import DecoderRuntime, GEMM_MODES
            mode = GEMM_MODES.get(self.gemm_mode, self.gemm_mode)
            if mode not in GEMM_MODES.values():
                raise ValueError('gemm_mode %r: expected one of %s' % (self.gemm_mode, sorted(GEMM_MODES)))
            if mode >= 2 and not self._split_gemm_covers(B * query_bbox.shape[1]):
                mode = 0                        # shapes outside the split kernels' coverage: the exact kernels (same arithmetic class)
            if self._runtime is None or self._runtime.gemm_mode != mode or self._runtime.overlap != self.overlap:
                self._runtime = DecoderRuntime(self, mode, self.overlap)
            if self.static_graph and query_bbox.dtype == torch.float32 and query_feat.dtype == torch.float32:
                # _finish (SparseBEVTransformer.forward): the outputs come back nan_to_num'ed in tensors of this call's own, written by
                # the step's last launch -- no torch kernel runs on the inference step
                out = self._runtime.step_graphs.run(query_bbox, query_feat, mlvl_feats, attn_mask, img_metas, finish=_finish)
                if out is not None:
                    if _finish:
                        return _Finished(out)
                    return out if _may_alias else (out[0].clone(), out[1].clone())
        ctx = DecoderContext(img_metas, B, query_bbox.device)
        query_bbox = query_bbox.float().contiguous()
        query_feat = query_feat.float().contiguous()
        if inference and not (layerwise or DUMP.enabled) and self._runtime.lazy_ok(mlvl_feats):
            # the eager step on the reference's NCHW lists (first sighting of a shape, graphs off, launch profiling): on-demand relayout
            # too -- only the feature units the sample points read are moved (runtime.forward_lazy; bit-identical to the dense pass)
            out = self._runtime.forward_lazy(query_bbox, query_feat, list(mlvl_feats), ctx, attn_mask, finish=_finish)[:2]
            return _Finished(out) if _finish else out
        feats = mlvl_feats if hasattr(mlvl_feats, 'levels') else FeaturePyramid(mlvl_feats)   # FeaturePyramid / cache.RingPyramid pass through
        if not inference:
            return self.forward_differentiable(query_bbox, query_feat, mlvl_feats, feats, attn_mask, ctx)
        if not (layerwise or DUMP.enabled):
            out = self._runtime.forward(query_bbox, query_feat, feats, ctx, attn_mask, finish=_finish)
            return _Finished(out) if _finish else out
        with torch.no_grad():
            return self._forward_layerwise(query_bbox, query_feat, feats, attn_mask, ctx)

    def invalidate_caches(self):
        """Call after weight writes that bypass autograd's version counter (``p.data.copy_(...)``: mmcv's Fp16OptimizerHook copying
        the fp32 master weights back, EMA hooks): drops the cached W^T of the backward and makes the inference runtime re-pack
        its weight images (row-chain pack, split-bf16 images) and drop its captured graphs.  In-place ops on the parameters
        themselves (every torch optimizer) are detected automatically."""
        AG.invalidate_caches()
        if self._runtime is not None:
            self._runtime._sig = ()         # never equal to a real signature: the next call re-binds (and clears the step graphs)

    def forward_differentiable(self, query_bbox, query_feat, mlvl_feats, feats, attn_mask, ctx):
        """Training / fine-tuning path (grad enabled and something requires grad): every op a HIP forward + HIP backward
        node; dropout only in train() mode.  Mirrors models/sparsebev_transformer.py:86-101 including the detach of the
        refined boxes between layers (:93)."""
        if hasattr(feats, 'frame_slots'):
            raise NotImplementedError('the online frame ring is an inference cache; train on [B, T*6, C, H, W] feature lists')
        if feats.levels[0].dtype != torch.float32:
            raise NotImplementedError('training needs fp32 feature maps (bf16 storage is an inference format)')
        orig = [f for f in mlvl_feats if torch.is_tensor(f)] if isinstance(mlvl_feats, (list, tuple)) else []
        token = AG.feature_token(feats, orig)       # ONE node hands the shared feature-gradient buffers to autograd (or None)
        layer = self.decoder_layer
        from .runtime import GEMM_MODES, GEMM_F16X3, GEMM_F16X4
        layer.train_gemm_f16 = GEMM_MODES.get(self.gemm_mode, self.gemm_mode) in (GEMM_F16X3, GEMM_F16X4)     # autograd.AdaptiveMixing
        saved = (layer.self_attn.attn_drop, layer.ffn_drop)
        if not self.training:
            layer.self_attn.attn_drop, layer.ffn_drop = 0.0, 0.0
        try:
            cls_scores, bbox_preds = [], []
            packed = layer.packed_train_weights()
            # the layers share their parameters: ONE gradient buffer per parameter and call, added to in the kernels' epilogues (AG.Tap)
            # (the packed q | k | v | tau and sampling tensors too: their gradient reaches the parameters through the one cat node each)
            tap = AG.Tap([p for p in layer.parameters()] + [t for t in packed if t.requires_grad]) if self.tap_param_grads else None
            for i in range(self.num_layers):
                query_feat, cls_score, bbox_pred = layer.forward_train(query_bbox, query_feat, feats, attn_mask, ctx, token, packed, tap)
                query_bbox = bbox_pred.detach()
                cls_scores.append(cls_score)
                bbox_preds.append(bbox_pred)
                if self.value_forcing is not None:      # tests only: evaluate the NEXT layer at recorded values, graph intact
                    ref_bbox, ref_feat = self.value_forcing
                    query_bbox = ref_bbox[i].to(query_bbox)
                    query_feat = query_feat + (ref_feat[i].to(query_feat) - query_feat).detach()
        finally:
            layer.self_attn.attn_drop, layer.ffn_drop = saved
        return torch.stack(cls_scores), torch.stack(bbox_preds)

    def _forward_layerwise(self, query_bbox, query_feat, feats, attn_mask, ctx):
        cls_scores, bbox_preds = [], []
        for i in range(self.num_layers):
            DUMP.stage_count = i
            query_feat, cls_score, bbox_pred = self.decoder_layer(query_bbox, query_feat, feats, attn_mask, ctx)
            query_bbox = bbox_pred.detach()
            cls_scores.append(cls_score)
            bbox_preds.append(bbox_pred)
        return torch.stack(cls_scores), torch.stack(bbox_preds)


@_register
class SparseBEVTransformer(_Base):
    """models/sparsebev_transformer.py:16-38."""

    def __init__(self, embed_dims, num_frames=8, num_points=4, num_layers=6, num_levels=4, num_classes=10,
                 code_size=10, pc_range=[], init_cfg=None):
        assert init_cfg is None, 'To prevent abnormal initialization behavior, init_cfg is not allowed to be set'
        super().__init__(init_cfg=init_cfg)
        self.embed_dims = embed_dims
        self.pc_range = pc_range
        self.decoder = SparseBEVTransformerDecoder(embed_dims, num_frames, num_points, num_layers, num_levels,
                                                   num_classes, code_size, pc_range=pc_range)

    @torch.no_grad()
    def init_weights(self):
        self.decoder.init_weights()

    def forward(self, query_bbox, query_feat, mlvl_feats, attn_mask, img_metas, layerwise=False):
        """Differentiable like the reference's module: with grad enabled and any input / parameter requiring grad the decoder
        runs its autograd path (HIP forward + HIP backward kernels, sparsebev_amd/autograd.py); otherwise the fused inference
        runtime.  train() additionally switches the dropouts on (mmcv's attn_drop / ffn_drop = 0.1)."""
        VERSION.require_supported()
        out = self.decoder(query_bbox, query_feat, mlvl_feats, attn_mask, img_metas, layerwise=layerwise, _may_alias=True, _finish=True)
        if isinstance(out, _Finished):      # the inference runtime: nan_to_num'ed by the step's own last launch (csrc/layout.hip::finish_outputs_kernel)
            return out[0], out[1]
        cls_scores, bbox_preds = out
        return torch.nan_to_num(cls_scores), torch.nan_to_num(bbox_preds)      # (training / layerwise / DUMP paths; out of place)
