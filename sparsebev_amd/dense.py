"""Dense half of the decoder layer (linears, LayerNorms, scale-adaptive self attention, adaptive mixing,
box refinement, NCHW->NHWC relayout) on the device.

``NATIVE`` records which of these are hand-written gfx950 kernels in libsbev_hip.so and which still run as
stock PyTorch-ROCm device ops (rocBLAS / ATen).  Every function here requires device tensors -- there is
no CPU path in the product.
"""
import math

import torch
import torch.nn.functional as F

# op -> implementation currently used on the device
NATIVE = {
    'linear': 'aten',
    'layer_norm': 'aten',
    'linear_ln_relu': 'aten',
    'self_attention': 'aten',
    'adaptive_mixing': 'aten',
    'refine_bbox': 'aten',
    'to_channels_last': 'aten',
}


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('sparsebev_amd.dense needs device tensors (no CPU fallback)')


def linear(x, w, b, relu=False, residual=None):
    _dev(x, w)
    y = F.linear(x, w, b)
    if relu:
        y = torch.relu(y)
    if residual is not None:
        y = y + residual
    return y


def layer_norm(x, w, b, eps=1e-5):
    _dev(x)
    return F.layer_norm(x, [x.shape[-1]], w, b, eps)


def linear_ln_relu(x, w, b, lnw, lnb):
    return torch.relu(layer_norm(linear(x, w, b), lnw, lnb))


def scale_adaptive_self_attention(query_bbox, x, pc_range, num_heads, in_w, in_b, out_w, out_b, tau_w, tau_b, pre_attn_mask=None):
    """models/sparsebev_transformer.py:210-228,236-248 + mmcv MultiheadAttention(batch_first) = x + MHA(x)."""
    _dev(query_bbox, x)
    B, Q, D = x.shape
    hd = D // num_heads
    cx = query_bbox[..., 0] * (pc_range[3] - pc_range[0]) + pc_range[0]
    cy = query_bbox[..., 1] * (pc_range[4] - pc_range[1]) + pc_range[1]
    xy = torch.stack([cx, cy], -1)
    dist = -torch.norm(xy[:, :, None, :] - xy[:, None, :, :], dim=-1)
    tau = F.linear(x, tau_w, tau_b)
    bias = dist[:, None] * tau.permute(0, 2, 1)[..., None]
    if pre_attn_mask is not None:
        bias = bias.masked_fill(pre_attn_mask[None, None], float('-inf'))
    qkv = F.linear(x, in_w, in_b)
    q, k, v = (t.reshape(B, Q, num_heads, hd).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    att = torch.softmax(torch.matmul(q / math.sqrt(hd), k.transpose(-1, -2)) + bias, dim=-1)
    o = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(B, Q, D)
    return x + F.linear(o, out_w, out_b)


def adaptive_mixing(x, query, pg_w, pg_b, op_w, op_b, out_points):
    """models/sparsebev_transformer.py:351-381.  x [B,Q,G,Pin,C], query [B,Q,D] -> [B,Q,D]."""
    _dev(x, query)
    B, Q, G, Pin, C = x.shape
    gen = F.linear(query, pg_w, pg_b).reshape(B * Q, G, -1)
    M = gen[..., : C * C].reshape(B * Q, G, C, C)
    S = gen[..., C * C:].reshape(B * Q, G, out_points, Pin)
    y = torch.relu(F.layer_norm(torch.matmul(x.reshape(B * Q, G, Pin, C), M), [Pin, C]))
    y = torch.relu(F.layer_norm(torch.matmul(S, y), [out_points, C]))
    return query + F.linear(y.reshape(B, Q, -1), op_w, op_b)


def refine_bbox(query_bbox, reg, vel_div):
    """refine_bbox + velocity / time_diff (models/sparsebev_transformer.py:155-160,179-183; inverse_sigmoid
    models/utils.py:87-102)."""
    _dev(query_bbox, reg)
    p = query_bbox[..., 0:3].clamp(0, 1)
    logit = torch.log(p.clamp(min=1e-5) / (1 - p).clamp(min=1e-5))
    xyz = torch.sigmoid(reg[..., 0:3] + logit)
    vel = reg[..., 8:]
    if vel_div is not None:
        vel = vel / vel_div[:, None, None]
    return torch.cat([xyz, reg[..., 3:8], vel], dim=-1)


def to_channels_last(f):
    """[B,TN,GC,H,W] -> contiguous [B,TN,H,W,GC]."""
    _dev(f)
    return f.permute(0, 1, 3, 4, 2).contiguous()
