"""Seeded synthetic inputs for the SparseBEV decoder hot path.

No dataset or checkpoint is reachable, so tests, ``smoke()`` and ``bench.py`` all draw from here:
a 6-camera ring rig (``lidar2img`` in the layout ``loaders/nuscenes_dataset.py:64-76`` produces),
sweep timestamps (``loaders/pipelines/loading.py:46``), head-style query initialisation
(``models/sparsebev_head.py:49-64``) and FPN-shaped feature pyramids.  numpy / torch only.
"""
import math

import numpy as np
import torch

PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]          # configs/r50_nuimg_704x256.py:8
N_VIEWS = 6

# name -> (image_h, image_w, [(H_l, W_l), ...]); sizes from SURVEY.md section 8 config table
PYRAMIDS = {
    'r50_704x256': (256, 704, [(64, 176), (32, 88), (16, 44), (8, 22)]),
    'r101_1408x512': (512, 1408, [(128, 352), (64, 176), (32, 88), (16, 44), (8, 22)]),
    'eva02_1600x640': (640, 1600, [(160, 400), (80, 200), (40, 100), (20, 50), (10, 25)]),
    'tiny': (256, 704, [(8, 22), (4, 11), (2, 6), (1, 3)]),
    'tiny5': (256, 704, [(16, 44), (8, 22), (4, 11), (2, 6), (1, 3)]),
}


def camera_rig(T, image_h, image_w):
    """lidar2img for T frames x 6 views, float64 [T*6,4,4], image index = t*6 + view.

    View i looks along lidar yaw 2*pi*i/6 (+0.01 rad per frame of ego rotation); lidar axes are
    x forward / y left / z up, camera axes x right / y down / z forward; focal 560 px at 704 px width
    (64 deg horizontal FOV, so neighbouring views overlap by ~4 deg); the camera sits 0.3 m right of,
    1.5 m above and 0.5 m behind the lidar origin along its own axes."""
    s = image_w / 704.0
    K = np.eye(4)
    K[0, 0] = K[1, 1] = 560.0 * s
    K[0, 2] = image_w / 2.0
    K[1, 2] = image_h / 2.0
    mats = []
    for t in range(T):
        for i in range(N_VIEWS):
            yaw = 2.0 * math.pi * i / N_VIEWS + 0.01 * t
            c, sn = math.cos(yaw), math.sin(yaw)
            R = np.array([[sn, -c, 0.0], [0.0, 0.0, -1.0], [c, sn, 0.0]])
            E = np.eye(4)
            E[:3, :3] = R
            E[:3, 3] = [0.3, 1.5, -0.5]
            mats.append(K @ E)
    return np.stack(mats)


def make_img_metas(B, T, image_h, image_w, frame_dt=0.5):
    """list[B] of the three meta fields the decoder reads (models/sparsebev_transformer.py:60-70,276)."""
    rig = camera_rig(T, image_h, image_w)
    metas = []
    for b in range(B):
        ts = [1.6e9 + 10.0 * b - frame_dt * (i // N_VIEWS) for i in range(T * N_VIEWS)]
        metas.append(dict(img_timestamp=ts,
                          lidar2img=[rig[i].copy() for i in range(T * N_VIEWS)],
                          img_shape=[(image_h, image_w, 3)] * (T * N_VIEWS)))
    return metas


def make_queries(B, Q, embed_dims=256, seed=0, z_norm=0.5):
    """query_bbox [B,Q,10], query_feat [B,Q,D] (fp32, CPU).  xy on the sqrt(Q) grid exactly as the head
    initialises them; z at `z_norm` of the range (-1 m: roughly ground level under the lidar -- with the
    head's literal z=0 -> -5 m nearly every near-range query projects below the image and the sampler
    would be benchmarked on mostly-skipped taps); log-dims w,l ~ N(0,0.25), log-h 0.5 (the head's 1.5 gives 4.5 m tall boxes),
    sin/cos N(0,1), velocity N(0,1) m/s so the temporal warp is exercised."""
    g = torch.Generator().manual_seed(seed)
    n = int(math.isqrt(Q))
    assert n * n == Q, 'num_query must be a perfect square (models/sparsebev_head.py:57-58)'
    ii, jj = torch.meshgrid(torch.arange(n), torch.arange(n), indexing='ij')
    xy = (torch.stack([ii, jj], dim=-1).reshape(Q, 2).float() + 0.5) / n
    bbox = torch.zeros(B, Q, 10)
    bbox[..., 0:2] = xy
    bbox[..., 2] = z_norm
    bbox[..., 3:5] = 0.5 * torch.randn(B, Q, 2, generator=g)
    bbox[..., 5] = 0.5
    bbox[..., 6:8] = torch.randn(B, Q, 2, generator=g)
    bbox[..., 8:10] = torch.randn(B, Q, 2, generator=g)
    feat = torch.randn(B, Q, embed_dims, generator=g)
    return bbox, feat


def make_features(B, T, level_sizes, channels=256, seed=0, device='cpu', dtype=torch.float32):
    """list[L] of [B, T*6, channels, H_l, W_l] i.i.d. N(0,1) -- the FPN output layout of
    models/sparsebev.py:126-131."""
    g = torch.Generator(device=device).manual_seed(seed)
    return [torch.randn(B, T * N_VIEWS, channels, h, w, generator=g, device=device, dtype=torch.float32).to(dtype)
            for (h, w) in level_sizes]


@torch.no_grad()
def randomize_zero_init(module_or_state, std=0.02, seed=0):
    """After init_weights() three generator weights are exactly zero (sampling_offset, parameter_generator,
    gen_tau: models/sparsebev_transformer.py:206-208,265-268,348-349), which would make offsets, dynamic
    mixing weights and tau query-independent.  Overwrite them with N(0, std^2)."""
    state = module_or_state if isinstance(module_or_state, dict) else dict(module_or_state.named_parameters())
    g = torch.Generator().manual_seed(seed)
    for name, p in state.items():
        if name.endswith(('sampling_offset.weight', 'parameter_generator.weight', 'gen_tau.weight')):
            p.copy_((std * torch.randn(p.shape, generator=g)).to(p.device, p.dtype))


def param_shapes(embed_dims=256, num_frames=8, num_points=4, num_levels=4, num_classes=10, code_size=10,
                 num_groups=4, num_heads=8, out_points=128, ffn_channels=512):
    """Names (reference state-dict keys minus the 'decoder.decoder_layer.' prefix)
    and shapes of the shared decoder layer's 48 tensors."""
    D, G = embed_dims, num_groups
    C = D // G
    pin = num_points * num_frames
    shp = {}

    def lin(name, o, i):
        shp[name + '.weight'] = (o, i)
        shp[name + '.bias'] = (o,)

    def ln(name):
        shp[name + '.weight'] = (D,)
        shp[name + '.bias'] = (D,)

    lin('position_encoder.0', D, 3); ln('position_encoder.1'); lin('position_encoder.3', D, D); ln('position_encoder.4')
    shp['self_attn.attention.attn.in_proj_weight'] = (3 * D, D)
    shp['self_attn.attention.attn.in_proj_bias'] = (3 * D,)
    lin('self_attn.attention.attn.out_proj', D, D)
    lin('self_attn.gen_tau', num_heads, D)
    lin('sampling.sampling_offset', G * num_points * 3, D)
    lin('sampling.scale_weights', G * num_points * num_levels, D)
    lin('mixing.parameter_generator', G * (C * C + pin * out_points), D)
    lin('mixing.out_proj', D, C * out_points * G)
    lin('ffn.layers.0.0', ffn_channels, D); lin('ffn.layers.1', D, ffn_channels)
    ln('norm1'); ln('norm2'); ln('norm3')
    lin('cls_branch.0', D, D); ln('cls_branch.1'); lin('cls_branch.3', D, D); ln('cls_branch.4')
    lin('cls_branch.6', num_classes, D)
    lin('reg_branch.0', D, D); lin('reg_branch.2', D, D); lin('reg_branch.4', code_size, D)
    return shp


def make_params(seed=0, **cfg):
    """Deterministic random decoder-layer parameters (CPU fp32 dict): PyTorch-default-like uniform
    linears, LayerNorm gains 1 + 0.1 N(0,1), and the reference's init_weights() conventions for the
    generator layers (uniform biases, models/sparsebev_transformer.py:146-153,206-208,265-268) with the
    zero-initialised weights replaced by N(0, 0.02^2)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(**cfg).items():
        is_ln = len(shape) == 1 and name.endswith('.weight')
        if is_ln:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(('sampling_offset.weight', 'parameter_generator.weight', 'gen_tau.weight')):
            t = 0.02 * torch.randn(shape, generator=g)
        elif name == 'sampling.sampling_offset.bias':
            t = torch.rand(shape, generator=g) - 0.5
        elif name == 'self_attn.gen_tau.bias':
            t = 2.0 * torch.rand(shape, generator=g)
        elif name == 'cls_branch.6.bias':
            t = torch.full(shape, -math.log(99.0))
        elif len(shape) == 2:
            bound = 1.0 / math.sqrt(shape[1])
            t = (2.0 * torch.rand(shape, generator=g) - 1.0) * bound
        else:
            t = 0.1 * torch.randn(shape, generator=g)
        out[name] = t
    return out


def checksum(tensors):
    """float64 sum of |x| over a tensor / list / dict of tensors -- stored in fixtures to prove that the
    seeded regeneration on another box reproduced the same inputs."""
    if isinstance(tensors, dict):
        tensors = [tensors[k] for k in sorted(tensors)]
    if torch.is_tensor(tensors):
        tensors = [tensors]
    return float(sum(t.detach().double().abs().sum().item() for t in tensors))


def grad_sample_indices(numel, n=8192, seed=1234):
    """The fixed subset of a big gradient tensor that fixture G11 keeps (flat indices, seeded); None = all of it."""
    if numel <= 70000:
        return None
    return torch.randperm(numel, generator=torch.Generator().manual_seed(seed + numel % 1000))[:n]
