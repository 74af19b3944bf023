"""CPU oracle for the SparseBEV sampling + mixing hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; the product (``sparsebev_amd``) never does and fails loudly without its HIP library.

This is a from-scratch restatement (functional style, one parameter dict) of the arithmetic of the
reference decoder.  Every function cites the reference lines it follows (paths relative to the
reference checkout).  Parity is PINNED: ``tests/golden/*.npz`` were produced by importing the
reference's own modules in the authoring container (``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` checks every function here against them.

Parameter naming: ``params`` is a ``dict[str, Tensor]`` whose keys are the reference state-dict
names with the prefix ``decoder.decoder_layer.`` stripped.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

N_VIEWS = 6          # models/sparsebev_sampling.py:45, models/sparsebev_transformer.py:75
N_GROUPS = 4         # models/sparsebev_transformer.py:123-124
N_HEADS = 8          # models/sparsebev_transformer.py:122
OUT_POINTS = 128     # models/sparsebev_transformer.py:124
EPS_HOMO = 1e-5      # models/sparsebev_sampling.py:27


# --------------------------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------------------------
def decode_bbox(bbox, pc_range):
    """models/bbox/utils.py:63-77 -> (xyz metres, wlh, yaw[...,1], vel)."""
    lo = bbox.new_tensor(pc_range[0:3])
    span = bbox.new_tensor([pc_range[3] - pc_range[0], pc_range[4] - pc_range[1], pc_range[5] - pc_range[2]])
    xyz = bbox[..., 0:3] * span + lo
    wlh = bbox[..., 3:6].exp()
    yaw = torch.atan2(bbox[..., 6:7], bbox[..., 7:8])
    vel = bbox[..., 8:10]
    return xyz, wlh, yaw, vel


def inverse_sigmoid(x, eps=1e-5):
    """models/utils.py:87-102."""
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


VERSION_NAME = 'v1.0.0'     # the reference's module-global VERSION.name (models/utils.py:320-325); tests may set 'v0.17.1'


def make_sample_points(query_bbox, offset, pc_range):
    """models/sparsebev_sampling.py:8-24 with models/utils.py:49-84 (rotation sign per VERSION: :66-77).

    query_bbox [B,Q,10]; offset [B,Q,GP,3] -> [B,Q,GP,3] metres."""
    xyz, wlh, yaw, _ = decode_bbox(query_bbox, pc_range)
    d = wlh[:, :, None, :] * offset
    c, s = torch.cos(yaw), torch.sin(yaw)                      # [B,Q,1]
    if VERSION_NAME == 'v0.17.1':
        s = -s
    dx = d[..., 0] * c + d[..., 1] * (-s)
    dy = d[..., 0] * s + d[..., 1] * c
    rotated = torch.stack([dx, dy, d[..., 2]], dim=-1)
    return xyz[:, :, None, :] + rotated


# --------------------------------------------------------------------------------------------
# A1 / A2: the multi-scale multi-view sampler
# --------------------------------------------------------------------------------------------
def msmv_sampling_gridsample(feats_cf, loc, weights):
    """A2 -- models/csrc/wrapper.py:14-38 (the reference's native-PyTorch sampler, the CPU baseline).

    feats_cf: list of [B',C,N,H,W]; loc [B',Q,P,3] in [0,1]; weights [B',Q,P,L] -> [B',Q,C,P]."""
    Bp, C = feats_cf[0].shape[:2]
    Q, P = loc.shape[1:3]
    grid = (loc * 2 - 1)[:, :, :, None, :]
    acc = torch.zeros(Bp, C, Q, P, dtype=feats_cf[0].dtype)
    for l, f in enumerate(feats_cf):
        s = F.grid_sample(f, grid, mode='bilinear', padding_mode='zeros', align_corners=True)[..., 0]
        acc = acc + s * weights[..., l].reshape(Bp, 1, Q, P)
    return acc.permute(0, 2, 1, 3)


def msmv_sampling_kernel_semantics(feats_cl, loc, weights):
    """A1 -- restatement of the CUDA kernel models/csrc/msmv_sampling/msmv_sampling_forward.cu:27-164
    (5-level variant :166-267 is the same loop with one more level).

    feats_cl: list of [B',N,H,W,C] channel-last; loc [B',Q,P,3]; weights [B',Q,P,L] -> [B',Q,C,P].
    view = round(z*(N-1)) (:109); h_im = y*(H-1), w_im = x*(W-1) (:123-124); a level contributes only
    if -1 < h_im < H and -1 < w_im < W (:126); each bilinear corner is zero outside the map (:47-66)."""
    Bp, N, _, _, C = feats_cl[0].shape
    Q, P = loc.shape[1:3]
    x, y = loc[..., 0], loc[..., 1]
    z = loc[..., 2] * (N - 1)
    view = torch.where(z >= 0, torch.floor(z + 0.5), torch.ceil(z - 0.5)).long()   # C round(): half away from zero
    view_ix = view.clamp(0, N - 1)
    b_ix = torch.arange(Bp)[:, None, None].expand(Bp, Q, P)
    acc = torch.zeros(Bp, Q, P, C, dtype=feats_cl[0].dtype)
    for l, f in enumerate(feats_cl):
        H, W = f.shape[2:4]
        h_im = y * (H - 1)
        w_im = x * (W - 1)
        ok = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h0 = torch.floor(h_im)
        w0 = torch.floor(w_im)
        lh, lw = h_im - h0, w_im - w0
        hh, hw = 1 - lh, 1 - lw
        h0, w0 = h0.long(), w0.long()
        val = torch.zeros_like(acc)
        for dh, dw, cw in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
            hc, wc = h0 + dh, w0 + dw
            inb = ok & (hc >= 0) & (hc <= H - 1) & (wc >= 0) & (wc <= W - 1)
            v = f[b_ix, view_ix, hc.clamp(0, H - 1), wc.clamp(0, W - 1)]          # [B',Q,P,C]
            # an out-of-map corner is never READ by the reference (:47-66 leave v1..v4 at 0): its value is 0, not "0 x whatever the
            # clamped pixel holds" -- the difference shows once a border pixel is Inf / NaN (fp16 backbones, val.py:115)
            v = torch.where(inb[..., None], v, torch.zeros_like(v))
            val = val + cw[..., None] * v
        acc = acc + val * weights[..., l][..., None]
    return acc.permute(0, 1, 3, 2).contiguous()


def msmv_sampling_backward(feats_cl, loc, weights, grad_out):
    """Backward of A1 -- restatement of models/csrc/msmv_sampling/msmv_sampling_backward.cu:29-224.

    feats_cl list of [B',N,H,W,C]; loc [B',Q,P,3]; weights [B',Q,P,L]; grad_out [B',Q,C,P].
    Returns (grad_feats list like feats_cl, grad_loc [B',Q,P,3] with a ZERO view component -- the CUDA op never
    writes it (:102-104) and the reference pipeline feeds a non-differentiable argmax there --, grad_weights)."""
    Bp, N, _, _, C = feats_cl[0].shape
    Q, P = loc.shape[1:3]
    g = grad_out.permute(0, 1, 3, 2)                                  # [B',Q,P,C]
    x, y = loc[..., 0], loc[..., 1]
    z = loc[..., 2] * (N - 1)
    view = torch.where(z >= 0, torch.floor(z + 0.5), torch.ceil(z - 0.5)).long().clamp(0, N - 1)
    b_ix = torch.arange(Bp)[:, None, None].expand(Bp, Q, P)
    grad_loc = torch.zeros_like(loc)
    grad_w = torch.zeros_like(weights)
    grad_feats = []
    for l, f in enumerate(feats_cl):
        H, W = f.shape[2:4]
        h_im, w_im = y * (H - 1), x * (W - 1)
        ok = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h0f, w0f = torch.floor(h_im), torch.floor(w_im)
        lh, lw = h_im - h0f, w_im - w0f
        hh, hw = 1 - lh, 1 - lw
        h0, w0 = h0f.long(), w0f.long()
        gf = torch.zeros_like(f)
        wl = weights[..., l]
        vals = []
        for dh, dw, cw in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
            hc, wc = h0 + dh, w0 + dw
            inb = ok & (hc >= 0) & (hc <= H - 1) & (wc >= 0) & (wc <= W - 1)
            hcc, wcc = hc.clamp(0, H - 1), wc.clamp(0, W - 1)
            v = torch.where(inb[..., None], f[b_ix, view, hcc, wcc], torch.zeros(1))
            vals.append(v)
            contrib = torch.where(inb, cw * wl, torch.zeros_like(cw))[..., None] * g
            gf.index_put_((b_ix, view, hcc, wcc), contrib, accumulate=True)
        v1, v2, v3, v4 = vals
        val = (hh * hw)[..., None] * v1 + (hh * lw)[..., None] * v2 + (lh * hw)[..., None] * v3 + (lh * lw)[..., None] * v4
        grad_w[..., l] = (g * val).sum(-1)
        gww = hh[..., None] * (v2 - v1) + lh[..., None] * (v4 - v3)        # d val / d w_im
        ghw = hw[..., None] * (v3 - v1) + lw[..., None] * (v4 - v2)        # d val / d h_im
        grad_loc[..., 0] += (W - 1) * wl * (g * gww).sum(-1)
        grad_loc[..., 1] += (H - 1) * wl * (g * ghw).sum(-1)
        grad_feats.append(gf)
    return grad_feats, grad_loc, grad_w


def regroup_features(mlvl_feats, channel_last):
    """models/sparsebev_transformer.py:73-85: [B,T*N,G*C,H,W] -> [B*T*G,N,H,W,C] or [B*T*G,C,N,H,W]."""
    out = []
    for f in mlvl_feats:
        B, TN, GC, H, W = f.shape
        T, C = TN // N_VIEWS, GC // N_GROUPS
        f = f.reshape(B, T, N_VIEWS, N_GROUPS, C, H, W)
        if channel_last:
            f = f.permute(0, 1, 3, 2, 5, 6, 4).reshape(B * T * N_GROUPS, N_VIEWS, H, W, C)
        else:
            f = f.permute(0, 1, 3, 4, 2, 5, 6).reshape(B * T * N_GROUPS, C, N_VIEWS, H, W)
        out.append(f.contiguous())
    return out


# --------------------------------------------------------------------------------------------
# A3 / A4: projection, hit mask, view selection
# --------------------------------------------------------------------------------------------
def project_points(sample_points, lidar2img, image_h, image_w, eps=EPS_HOMO):
    """A3 (i)-(iii) -- models/sparsebev_sampling.py:49-79, and the A4 DUMP tap (:82-86).

    sample_points [B,Q,T,GP,3]; lidar2img [B,T*N,4,4].
    Returns uvh [B,T,N,Q,GP,3] = (u/image_w, v/image_h, max(homo,eps)) and valid [B,T,N,Q,GP] (0/1 float).

    BIT-EXACT contract (SURVEY.md section 7 'Bit-exact hit mask'): the reference's batched fp32 4x4
    matmul equals ((m0*x + m1*y) + m2*z) + m3*1 with separate multiply and add roundings; the two
    divisions are IEEE divides.  torch CPU elementwise mul/add never fuse, so this is that order."""
    B, Q, T, GP, _ = sample_points.shape
    M = lidar2img.reshape(B, T, N_VIEWS, 1, 1, 4, 4)
    p = sample_points.permute(0, 2, 1, 3, 4)[:, :, None]       # [B,T,1,Q,GP,3]
    px, py, pz = p[..., 0], p[..., 1], p[..., 2]

    def row(r):
        return ((M[..., r, 0] * px + M[..., r, 1] * py) + M[..., r, 2] * pz) + M[..., r, 3] * 1.0

    uh, vh, homo = row(0), row(1), row(2)
    hn = torch.maximum(homo, torch.zeros_like(homo) + eps)
    u = (uh / hn) / image_w
    v = (vh / hn) / image_h
    valid = ((homo > eps) & (v > 0.0) & (v < 1.0) & (u > 0.0) & (u < 1.0)).float()
    return torch.stack([u, v, hn], dim=-1), valid


def select_view(uvh, valid):
    """A3 (iv) -- models/sparsebev_sampling.py:88-109: first hitting view (argmax of the 0/1 mask,
    0 when none hits); returns i_view [B,T,Q,GP] (int64) and loc [B,T,Q,GP,3] = (u, v, i_view/(N-1))."""
    i_view = torch.argmax(valid.permute(0, 1, 3, 4, 2), dim=-1)            # [B,T,Q,GP]
    uv = uvh[..., 0:2].permute(0, 1, 3, 4, 2, 5)                           # [B,T,Q,GP,N,2]
    sel = torch.gather(uv, 4, i_view[..., None, None].expand(*i_view.shape, 1, 2))[..., 0, :]
    loc = torch.cat([sel, (i_view.float() / (N_VIEWS - 1))[..., None]], dim=-1)
    return i_view, loc


def sampling_4d(sample_points, feats, scale_weights, lidar2img, image_h, image_w, sampler, eps=EPS_HOMO):
    """A3 -- models/sparsebev_sampling.py:27-130 end to end.

    sample_points [B,Q,T,G,P,3]; scale_weights [B,Q,G,T,P,L]; feats in whichever layout `sampler`
    wants (msmv_sampling_gridsample: channel-first; msmv_sampling_kernel_semantics: channel-last).
    Returns ([B,Q,G,T*P,C], taps) where taps = dict(uvh, valid, i_view, loc_bp, w_bp).

    Reference quirks kept on purpose:
      q1  points are flattened (b,t,g) but weights (b,g,t) (:112-119), so sample batch b'=(b*T+t)*G+g
          is weighted by scale_weights[b, :, g', t'] with g'*T + t' = t*G + g;
      q2  the hit mask is NOT applied to the output; no-hit points sample view 0."""
    B, Q, T, G, P, _ = sample_points.shape
    uvh, valid = project_points(sample_points.reshape(B, Q, T, G * P, 3), lidar2img, image_h, image_w, eps)
    i_view, loc = select_view(uvh, valid)                                   # [B,T,Q,GP,(3)]
    loc_bp = loc.reshape(B, T, Q, G, P, 3).permute(0, 1, 3, 2, 4, 5).reshape(B * T * G, Q, P, 3)
    L = scale_weights.shape[-1]
    w_bp = scale_weights.reshape(B, Q, G, T, P, L).permute(0, 2, 3, 1, 4, 5).reshape(B * G * T, Q, P, L)
    out = sampler(feats, loc_bp.contiguous(), w_bp.contiguous())            # [B',Q,C,P]
    C = out.shape[2]
    out = out.reshape(B, T, G, Q, C, P).permute(0, 3, 2, 1, 5, 4).reshape(B, Q, G, T * P, C)
    return out, dict(uvh=uvh, valid=valid, i_view=i_view, loc_bp=loc_bp, w_bp=w_bp)


# --------------------------------------------------------------------------------------------
# A5: adaptive spatio-temporal sampling front end
# --------------------------------------------------------------------------------------------
def sampling_front(params, query_bbox, query_feat, time_diff, pc_range, T, P, L):
    """A5 -- models/sparsebev_transformer.py:270-300: sample points [B,Q,T,G,P,3] and softmaxed
    scale weights [B,Q,G,T,P,L] (expanded over T)."""
    B, Q = query_bbox.shape[:2]
    G = N_GROUPS
    off = F.linear(query_feat, params['sampling.sampling_offset.weight'], params['sampling.sampling_offset.bias'])
    pts = make_sample_points(query_bbox, off.view(B, Q, G * P, 3), pc_range)          # [B,Q,GP,3]
    pts = pts.reshape(B, Q, 1, G, P, 3).expand(B, Q, T, G, P, 3)
    shift = query_bbox[..., 8:10].detach()[:, :, None, :] * time_diff[:, None, :, None]   # [B,Q,T,2]; vel detached (:288)
    pts = torch.cat([pts[..., 0:2] - shift[:, :, :, None, None, :], pts[..., 2:3]], dim=-1)
    sw = F.linear(query_feat, params['sampling.scale_weights.weight'], params['sampling.scale_weights.bias'])
    sw = torch.softmax(sw.view(B, Q, G, 1, P, L), dim=-1).expand(B, Q, G, T, P, L)
    return pts, sw


# --------------------------------------------------------------------------------------------
# A6: adaptive mixing
# --------------------------------------------------------------------------------------------
def adaptive_mixing(params, x, query):
    """A6 -- models/sparsebev_transformer.py:351-381.  x [B,Q,G,Pin,C]; query [B,Q,D] -> [B,Q,D]."""
    B, Q, G, Pin, C = x.shape
    gen = F.linear(query, params['mixing.parameter_generator.weight'], params['mixing.parameter_generator.bias'])
    gen = gen.reshape(B * Q, G, -1)
    Pout = (gen.shape[-1] - C * C) // Pin
    M = gen[..., : C * C].reshape(B * Q, G, C, C)
    S = gen[..., C * C:].reshape(B * Q, G, Pout, Pin)
    y = torch.matmul(x.reshape(B * Q, G, Pin, C), M)
    y = torch.relu(F.layer_norm(y, [Pin, C]))
    y = torch.matmul(S, y)
    y = torch.relu(F.layer_norm(y, [Pout, C]))
    y = F.linear(y.reshape(B, Q, -1), params['mixing.out_proj.weight'], params['mixing.out_proj.bias'])
    return query + y


# --------------------------------------------------------------------------------------------
# A7: scale-adaptive self attention
# --------------------------------------------------------------------------------------------
def self_attention(params, query_bbox, query_feat, pc_range, pre_attn_mask=None):
    """A7 -- models/sparsebev_transformer.py:210-228,236-248 + mmcv 1.6.0 MultiheadAttention
    (batch_first, identity + attn; wraps torch.nn.MultiheadAttention -- restated here explicitly)."""
    B, Q, D = query_feat.shape
    hd = D // N_HEADS
    xy = decode_bbox(query_bbox.detach(), pc_range)[0][..., :2]                      # calc_bbox_dists is @torch.no_grad (:236)
    dist = -torch.norm(xy[:, :, None, :] - xy[:, None, :, :], dim=-1)               # [B,Q,Q]
    tau = F.linear(query_feat, params['self_attn.gen_tau.weight'], params['self_attn.gen_tau.bias'])
    bias = dist[:, None] * tau.permute(0, 2, 1)[..., None]                           # [B,H,Q,Q], tau indexed by row
    if pre_attn_mask is not None:
        bias = bias.masked_fill(pre_attn_mask[None, None], float('-inf'))
    qkv = F.linear(query_feat, params['self_attn.attention.attn.in_proj_weight'],
                   params['self_attn.attention.attn.in_proj_bias'])
    q, k, v = (t.reshape(B, Q, N_HEADS, hd).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    logits = torch.matmul(q / math.sqrt(hd), k.transpose(-1, -2)) + bias
    att = torch.matmul(torch.softmax(logits, dim=-1), v).permute(0, 2, 1, 3).reshape(B, Q, D)
    att = F.linear(att, params['self_attn.attention.attn.out_proj.weight'],
                   params['self_attn.attention.attn.out_proj.bias'])
    return query_feat + att


# --------------------------------------------------------------------------------------------
# A8 / A9: decoder layer and decoder
# --------------------------------------------------------------------------------------------
def _ln(params, name, x):
    return F.layer_norm(x, [x.shape[-1]], params[name + '.weight'], params[name + '.bias'])


def _lin(params, name, x):
    return F.linear(x, params[name + '.weight'], params[name + '.bias'])


def time_diff_from_metas(img_metas, B):
    """A9 -- models/sparsebev_transformer.py:60-64: float64 mean over the 6 cameras, then fp32."""
    ts = np.array([m['img_timestamp'] for m in img_metas], dtype=np.float64).reshape(B, -1, N_VIEWS)
    return torch.from_numpy(np.mean(ts[:, :1, :] - ts, axis=-1).astype(np.float32))


def decoder_layer(params, query_bbox, query_feat, feats, time_diff, lidar2img, image_h, image_w,
                  pc_range, T, P, L, sampler, pre_attn_mask=None, taps=None):
    """A8 -- models/sparsebev_transformer.py:162-193."""
    pos = query_bbox[..., :3]
    pos = torch.relu(_ln(params, 'position_encoder.1', _lin(params, 'position_encoder.0', pos)))
    pos = torch.relu(_ln(params, 'position_encoder.4', _lin(params, 'position_encoder.3', pos)))
    x = query_feat + pos
    x = _ln(params, 'norm1', self_attention(params, query_bbox, x, pc_range, pre_attn_mask))
    pts, sw = sampling_front(params, query_bbox, x, time_diff, pc_range, T, P, L)
    sampled, t = sampling_4d(pts, feats, sw, lidar2img, image_h, image_w, sampler)
    if taps is not None:
        taps.append(t)
    x = _ln(params, 'norm2', adaptive_mixing(params, sampled, x))
    ffn = _lin(params, 'ffn.layers.1', torch.relu(_lin(params, 'ffn.layers.0.0', x)))
    x = _ln(params, 'norm3', x + ffn)
    c = torch.relu(_ln(params, 'cls_branch.1', _lin(params, 'cls_branch.0', x)))
    c = torch.relu(_ln(params, 'cls_branch.4', _lin(params, 'cls_branch.3', c)))
    cls = _lin(params, 'cls_branch.6', c)
    r = torch.relu(_lin(params, 'reg_branch.0', x))
    r = torch.relu(_lin(params, 'reg_branch.2', r))
    reg = _lin(params, 'reg_branch.4', r)
    xyz = torch.sigmoid(reg[..., 0:3] + inverse_sigmoid(query_bbox[..., 0:3]))     # refine_bbox :155-160
    bbox = torch.cat([xyz, reg[..., 3:]], dim=-1)
    if time_diff.shape[1] > 1:                                                     # :179-183
        td = time_diff.clone()
        td[td < 1e-5] = 1.0
        bbox = torch.cat([bbox[..., :8], bbox[..., 8:] / td[:, 1:2, None]], dim=-1)
    return x, cls, bbox


def decoder_prologue(mlvl_feats, img_metas, sampler):
    """A9 -- models/sparsebev_transformer.py:60-85: time_diff, lidar2img tensor, feature regroup.
    Does not mutate its inputs (the reference does)."""
    B = mlvl_feats[0].shape[0]
    time_diff = time_diff_from_metas(img_metas, B)
    lidar2img = torch.from_numpy(np.asarray([m['lidar2img'] for m in img_metas]).astype(np.float32))
    image_h, image_w = img_metas[0]['img_shape'][0][:2]
    feats = regroup_features(mlvl_feats, channel_last=(sampler is msmv_sampling_kernel_semantics))
    return feats, time_diff, lidar2img, image_h, image_w


def decoder(params, query_bbox, query_feat, mlvl_feats, img_metas, pc_range, num_layers=6,
            num_points=4, sampler=None, pre_attn_mask=None, taps=None, forced_inputs=None):
    """A9 -- models/sparsebev_transformer.py:32-38,56-101 (inference; shared layer weights).

    mlvl_feats: list[L] of [B,T*N,G*C,H,W]; img_metas: list[B] of dict(img_timestamp, lidar2img, img_shape).
    Returns cls [layers,B,Q,classes], bbox [layers,B,Q,10], feat [layers,B,Q,D].
    forced_inputs: optional list of (query_bbox, query_feat) per layer (teacher forcing for parity
    tests: a random-init 6-layer decoder amplifies fp32 rounding noise ~5x per layer)."""
    sampler = sampler or msmv_sampling_gridsample
    T = mlvl_feats[0].shape[1] // N_VIEWS
    feats, time_diff, lidar2img, image_h, image_w = decoder_prologue(mlvl_feats, img_metas, sampler)
    cls_all, box_all, feat_all = [], [], []
    for i in range(num_layers):
        if forced_inputs is not None:
            query_bbox, query_feat = forced_inputs[i]
        query_feat, cls, bbox = decoder_layer(params, query_bbox, query_feat, feats, time_diff, lidar2img,
                                              image_h, image_w, pc_range, T, num_points, len(mlvl_feats),
                                              sampler, pre_attn_mask, taps)
        query_bbox = bbox.detach().clone()
        cls_all.append(cls)
        box_all.append(bbox)
        feat_all.append(query_feat)
    return torch.nan_to_num(torch.stack(cls_all)), torch.nan_to_num(torch.stack(box_all)), torch.stack(feat_all)


# --------------------------------------------------------------------------------------------
# seeded synthetic inputs shared by tests, smoke() and bench.py
# --------------------------------------------------------------------------------------------
def strip_prefix(state_dict, prefix='decoder.decoder_layer.'):
    return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------------------------
# detection-head pre / post-processing
# --------------------------------------------------------------------------------------------
def head_prepare(init_query_bbox, label_enc_weight, num_classes, B):
    """models/sparsebev_head.py:70 + the eval branch of prepare_for_dn_input (:123-126,209-211).
    init_query_bbox [Q,10], label_enc_weight [num_classes+1, D-1] -> query_bbox [B,Q,10], query_feat [B,Q,D]."""
    Q = init_query_bbox.shape[0]
    feat = torch.cat([label_enc_weight[num_classes].repeat(Q, 1), torch.zeros(Q, 1)], dim=1)
    return init_query_bbox.clone().repeat(B, 1, 1), feat.repeat(B, 1, 1)


def head_postprocess(bbox_preds, pc_range):
    """models/sparsebev_head.py:85-95: centre back to metres (separate fp32 mul and add) and the head's column order
    (cx, cy, w, l, cz, h, sin, cos, vx, vy).  bbox_preds [..., 10] as the decoder returns them."""
    b = bbox_preds.clone()
    b[..., 0] = b[..., 0] * (pc_range[3] - pc_range[0]) + pc_range[0]
    b[..., 1] = b[..., 1] * (pc_range[4] - pc_range[1]) + pc_range[1]
    b[..., 2] = b[..., 2] * (pc_range[5] - pc_range[2]) + pc_range[2]
    return torch.cat([b[..., 0:2], b[..., 3:5], b[..., 2:3], b[..., 5:10]], dim=-1)


def denormalize_bbox(nb):
    """models/bbox/utils.py:26-47: (cx, cy, w, l, cz, h, sin, cos, vx, vy) -> (cx, cy, cz, w, l, h, rot, vx, vy)."""
    rot = torch.atan2(nb[..., 6:7], nb[..., 7:8])
    return torch.cat([nb[..., 0:2], nb[..., 4:5], nb[..., 2:4].exp(), nb[..., 5:6].exp(), rot, nb[..., 8:10]], dim=-1)


def nms_free_decode_single(cls_scores, bbox_preds, num_classes, max_num, score_threshold, post_center_range):
    """models/bbox/coders/nms_free_coder.py:37-88.  cls_scores [Q,NC] logits, bbox_preds [Q,10] head format.
    Ties in the top-k are resolved by (score desc, flat index asc) -- torch.topk leaves them unspecified."""
    s = cls_scores.sigmoid().reshape(-1)
    order = sorted(range(s.numel()), key=lambda i: (-float(s[i]), i))[:max_num] if s.numel() <= 4096 else None
    if order is None:                       # large inputs: stable argsort on the logits (same total order unless
        key = cls_scores.reshape(-1)        # distinct logits collapse to one fp32 sigmoid value)
        order = torch.argsort(-key.double(), stable=True)[:max_num].tolist()
    idx = torch.tensor(order, dtype=torch.long)
    scores = s[idx]
    labels = idx % num_classes
    boxes = denormalize_bbox(bbox_preds[torch.div(idx, num_classes, rounding_mode='trunc')])
    limit = torch.tensor(post_center_range, dtype=torch.float32)
    mask = (boxes[..., :3] >= limit[:3]).all(1) & (boxes[..., :3] <= limit[3:]).all(1)
    if score_threshold:
        mask &= scores > score_threshold
    return {'bboxes': boxes[mask], 'scores': scores[mask], 'labels': labels[mask]}


def nms_free_decode(all_cls_scores, all_bbox_preds, num_classes, max_num, score_threshold, post_center_range):
    """models/bbox/coders/nms_free_coder.py:90-111: last decoder layer, one dict per sample."""
    cls, box = all_cls_scores[-1], all_bbox_preds[-1]
    return [nms_free_decode_single(cls[i], box[i], num_classes, max_num, score_threshold, post_center_range)
            for i in range(cls.shape[0])]


def get_bboxes(decoded):
    """models/sparsebev_head.py:463-482: gravity centre -> bottom centre (+ the 'v0.17.1' w/l swap and yaw flip); returns
    [boxes [n,9], scores, labels] per sample (the reference wraps boxes in LiDARInstance3DBoxes(bboxes, 9))."""
    out = []
    for d in decoded:
        b = d['bboxes'].clone()
        b[:, 2] = b[:, 2] - b[:, 5] * 0.5
        if VERSION_NAME == 'v0.17.1':                           # :472-476
            w, l = b[:, 3].clone(), b[:, 4].clone()
            b[:, 3], b[:, 4] = l, w
            b[:, 6] = -b[:, 6] - math.pi / 2
        out.append([b, d['scores'], d['labels']])
    return out
