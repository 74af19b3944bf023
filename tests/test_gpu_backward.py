"""GPU parity of the decoder's BACKWARD kernels: every autograd.Function of
sparsebev_amd.autograd against torch autograd in fp64 on the CPU for the same op, the whole drop-in module against
fixture G11 = the REFERENCE decoder's own autograd gradients (tests/golden/make_golden.py::main_train), and the
training-mode dropouts against a host re-statement of their counter-based masks."""
import copy
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from sparsebev_amd import _lib, autograd as AG, synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer, FeaturePyramid, DecoderContext

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PREFIX = 'decoder.decoder_layer.'


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


@pytest.mark.parametrize('M,N,K', [(256, 32768, 900), (32768, 256, 900), (260, 16384, 37), (4096, 1028, 64)])
@pytest.mark.parametrize('wide', [False, True])
def test_gemm_tn_f16s_vs_fp64_not_worse_than_the_exact_kernel(M, N, K, wide):
    """grad_W-shaped product on the fp16 hi + lo kernel (sbev_gemm_tn_f16s): error against fp64 <= the exact f32-MFMA kernel's
    (sbev_gemm_f32, same layouts) -- also with operands spread over 12 binades -- ragged M / N / K, accumulation into a strided
    destination, and scales from sbev_f16s_tensor_scale."""
    from sparsebev_amd import dense
    g = torch.Generator().manual_seed(M + N + K)
    A, Bm = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g) / K ** 0.5
    if wide:
        A = A * torch.exp2(torch.randint(-12, 1, A.shape, generator=g).float())
        Bm = Bm * torch.exp2(torch.randint(-12, 1, Bm.shape, generator=g).float())
    ref = A.double().t() @ Bm.double()
    Ad, Bd = A.to(DEV), Bm.to(DEV)
    sa, sb = dense.f16s_tensor_scale(Ad), dense.f16s_tensor_scale(Bd)
    for t, sc in ((A, sa), (Bm, sb)):
        up, down = sc.cpu().tolist()
        assert up * down == 1.0 and 2.0 ** 14 <= t.abs().max().item() * up < 2.0 ** 15
    out = dense.gemm_tn_f16s(Ad, M, sa, Bd, N, sb, M, N, K)
    exact = AG.gemm(Ad, True, M, Bd, True, N, M, N, K)
    e16 = (out.cpu().double() - ref).abs()
    e32 = (exact.cpu().double() - ref).abs()
    assert e16.max() <= 1.05 * e32.max() + 1e-12, (e16.max().item(), e32.max().item())
    assert e16.pow(2).mean().sqrt() <= 1.05 * e32.pow(2).mean().sqrt() + 1e-12
    assert e16.max() < 3e-6 * max(1.0, ref.abs().max().item())
    # accumulate into a strided destination; a caller-side bound instead of the measured scale (any 2^e with max |x| 2^e < 65504)
    C0 = torch.randn(M, N + 4, generator=g)
    dst = C0.to(DEV)
    e = int(torch.log2(60000.0 / A.abs().max()).floor().item())
    dense.gemm_tn_f16s(Ad, M, torch.tensor([2.0 ** e, 2.0 ** -e], device=DEV), Bd, N, sb, M, N, K, out=dst, ldc=N + 4, accumulate=True)
    ref2 = C0.double()
    ref2[:, :N] += ref
    assert (dst.cpu().double() - ref2).abs().max() < 3e-6 * max(1.0, ref.abs().max().item()) + 1e-6
    assert torch.equal(dst[:, N:].cpu(), C0[:, N:])


def test_mixing_backward_item_maxima_are_the_maxima_of_grad_params():
    g = torch.Generator().manual_seed(5)
    BQ, G, Pin, C, Pout = 37, 4, 32, 64, 128
    NP = C * C + Pout * Pin
    x = torch.randn(BQ, G, Pin, C, generator=g).to(DEV)
    params = (torch.randn(BQ, G * NP, generator=g) * 0.2).to(DEV)
    gy = (torch.randn(BQ, G * Pout * C, generator=g) * torch.exp2(torch.randint(-6, 6, (BQ, 1), generator=g).float())).to(DEV)
    lib = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    gx0, gp0 = torch.empty_like(x), torch.empty_like(params)
    gx1, gp1 = torch.empty_like(x), torch.empty_like(params)
    imax = torch.empty(BQ * G * 4, device=DEV)           # four partial maxima per item
    assert lib.sbev_adaptive_mixing_bwd_f32(p(x), p(params), p(gy), p(gx0), p(gp0), BQ, G, Pin, C, Pout, 1e-5, None) == 0
    assert lib.sbev_adaptive_mixing_bwd_max_f32(p(x), p(params), p(gy), p(gx1), p(gp1), p(imax), BQ, G, Pin, C, Pout, 1e-5, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(gx0, gx1) and torch.equal(gp0, gp1)
    assert torch.equal(imax.reshape(BQ * G, 4).amax(dim=1), gp1.reshape(BQ * G, NP).abs().amax(dim=1))


@pytest.mark.parametrize('M,N,K', [(900, 256, 256), (900, 256, 32768), (256, 32768, 900), (32768, 256, 900), (900, 32768, 256),
                                   (37, 10, 256), (256, 3, 72), (130, 129, 33), (1, 5, 7), (3600, 256, 8192)])
def test_gemm_any_all_layouts_vs_fp64(M, N, K):
    """sbev_gemm_f32 for the four operand layouts, ragged M / N / K (zero-filled tiles), the element-wise staging path
    (ld % 4 != 0) and the split-K plan (few tiles, long K)."""
    g = torch.Generator().manual_seed(M + N + K)
    for ak in (False, True):
        for bk in (False, True):
            A = torch.randn((K, M) if ak else (M, K), generator=g)
            Bm = torch.randn((K, N) if bk else (N, K), generator=g) / K ** 0.5
            ref = (A.double().t() if ak else A.double()) @ (Bm.double() if bk else Bm.double().t())
            out = AG.gemm(A.to(DEV), ak, A.shape[1], Bm.to(DEV), bk, Bm.shape[1], M, N, K)
            assert (out.cpu().double() - ref).abs().max() < 3e-5 * max(1.0, ref.abs().max().item()), (ak, bk)
    # accumulate into a strided destination
    A, Bm = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) / K ** 0.5
    C0 = torch.randn(M, N + 3, generator=g)
    out = C0.to(DEV)
    AG.gemm(A.to(DEV), False, K, Bm.to(DEV), True, N, M, N, K, out=out, ldc=N + 3, accumulate=True)
    ref = C0.double()
    ref[:, :N] += A.double() @ Bm.double()
    assert (out.cpu().double() - ref).abs().max() < 3e-5 * max(1.0, ref.abs().max().item())


@torch.enable_grad()
@pytest.mark.parametrize('M,N,K,relu,res', [(900, 256, 256, True, False), (900, 256, 512, False, True), (37, 10, 256, False, False),
                                             (900, 776, 256, False, False), (64, 32768, 256, False, False), (64, 256, 32768, False, True)])
def test_linear_function_vs_torch(M, N, K, relu, res):
    g = torch.Generator().manual_seed(M + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    gy = torch.randn(M, N, generator=g)
    dv = [t.to(DEV).requires_grad_(True) for t in (x, w, b)] + ([r.to(DEV).requires_grad_(True)] if res else [None])
    y = AG.linear(dv[0], dv[1], dv[2], relu=relu, residual=dv[3])
    y.backward(gy.to(DEV))
    cv = [t.double().requires_grad_(True) for t in (x, w, b)] + ([r.double().requires_grad_(True)] if res else [None])
    yc = F.linear(cv[0], cv[1], cv[2])
    yc = yc.relu() if relu else yc
    yc = yc + cv[3] if res else yc
    yc.backward(gy.double())
    assert rel(y, yc) < 1e-5
    for d, c in zip(dv, cv):
        if d is not None:
            assert rel(d.grad, c.grad) < 2e-5


@torch.enable_grad()
@pytest.mark.parametrize('M,N,relu,add', [(900, 256, False, False), (900, 256, True, True), (37, 512, True, False), (3600, 256, False, True)])
def test_layer_norm_function_vs_torch(M, N, relu, add):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g) * 2 + 0.3
    w, b = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    a = torch.randn(M, N, generator=g) if add else None
    gy = torch.randn(M, N, generator=g)
    dv = [t.to(DEV).requires_grad_(True) for t in (x, w, b)] + ([a.to(DEV).requires_grad_(True)] if add else [None])
    y = AG.layer_norm(dv[0], dv[1], dv[2], relu=relu, add_after=dv[3])
    y.backward(gy.to(DEV))
    cv = [t.double().requires_grad_(True) for t in (x, w, b)] + ([a.double().requires_grad_(True)] if add else [None])
    yc = F.layer_norm(cv[0], [N], cv[1], cv[2])
    yc = yc.relu() if relu else yc
    yc = yc + cv[3] if add else yc
    yc.backward(gy.double())
    for d, c in zip(dv, cv):
        if d is not None:
            assert rel(d.grad, c.grad) < 2e-5


@torch.enable_grad()
def test_position_encoder_first_stage_function_vs_torch():
    g = torch.Generator().manual_seed(5)
    B, Q, D = 2, 450, 256
    bbox = torch.rand(B, Q, 10, generator=g)
    w, b = torch.randn(D, 3, generator=g), torch.randn(D, generator=g)
    lw, lb = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    gy = torch.randn(B, Q, D, generator=g)
    dv = [t.to(DEV).requires_grad_(True) for t in (bbox, w, b, lw, lb)]
    y = AG.Linear3LnRelu.apply(*dv)
    y.backward(gy.to(DEV))
    cv = [t.double().requires_grad_(True) for t in (bbox, w, b, lw, lb)]
    yc = F.layer_norm(F.linear(cv[0][..., :3], cv[1], cv[2]), [D], cv[3], cv[4]).relu()
    yc.backward(gy.double())
    assert rel(y, yc) < 1e-5
    for d, c in zip(dv, cv):
        assert rel(d.grad, c.grad) < 2e-5
    assert dv[0].grad[..., 3:].abs().max() == 0


def _mix32(z):
    z = (z + np.uint64(0x9e3779b97f4a7c15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
    return ((z ^ (z >> np.uint64(31))) >> np.uint64(32)).astype(np.uint32)


def _keep_mask(seed, p, shape_bhqq):
    """Host re-statement of the kernels' counter-based dropout: keep(i) = mix32(seed * 0x100000001b3 + i) >= p * 2^32."""
    n = int(np.prod(shape_bhqq))
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001b3)
        keep = _mix32(idx) >= np.uint32(int(float(np.float32(p)) * 4294967296.0))
    return torch.from_numpy(keep.reshape(shape_bhqq))


def _sasa_ref(qkvt, bbox, mask, H, keep=None, p=0.0):
    """fp64 dense restatement of the attention core on the packed q | k | v | tau rows."""
    B, Q, _ = qkvt.shape
    D = H * 32
    q, k, v, tau = qkvt[..., :D], qkvt[..., D:2 * D], qkvt[..., 2 * D:3 * D], qkvt[..., 3 * D:3 * D + H]
    q = q.reshape(B, Q, H, 32).transpose(1, 2) / 32 ** 0.5
    k = k.reshape(B, Q, H, 32).transpose(1, 2)
    v = v.reshape(B, Q, H, 32).transpose(1, 2)
    lo = torch.tensor(S.PC_RANGE[:2], dtype=torch.float32)
    span = torch.tensor([S.PC_RANGE[3] - S.PC_RANGE[0], S.PC_RANGE[4] - S.PC_RANGE[1]], dtype=torch.float32)
    c = (bbox[..., :2] * span + lo).double()
    dist = (c[:, :, None] - c[:, None]).norm(dim=-1)                        # [B,Q,Q]
    s = q @ k.transpose(-1, -2) - dist[:, None] * tau.transpose(1, 2)[..., None]
    if mask is not None:
        s = s.masked_fill(mask[None, None], float('-inf'))
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = pr * keep.double() / (1 - p)
    return (pr @ v).transpose(1, 2).reshape(B, Q, D)


@torch.enable_grad()
@pytest.mark.parametrize('Q,use_mask,p', [(100, False, 0.0), (37, True, 0.0), (130, False, 0.1), (64, True, 0.25)])
def test_sasa_core_function_vs_fp64(Q, use_mask, p):
    g = torch.Generator().manual_seed(Q)
    B, H = 2, 8
    D = H * 32
    qkvt = torch.randn(B, Q, 3 * D + H, generator=g) * 0.7
    qkvt[..., 3 * D:] = torch.rand(B, Q, H, generator=g) * 0.2
    bbox, _ = S.make_queries(B, 36, seed=Q)
    bbox = bbox[:, :1].expand(B, Q, 10).clone()
    bbox[..., :2] = torch.rand(B, Q, 2, generator=g)
    mask = None
    if use_mask:
        mask = torch.zeros(Q, Q, dtype=torch.bool)
        mask[:10, 10:] = True
        mask[10:, :4] = True
    gy = torch.randn(B, Q, D, generator=g)
    seed = 987654321
    keep = _keep_mask(seed, p, (B, H, Q, Q)) if p > 0 else None
    qd = qkvt.to(DEV).requires_grad_(True)
    md = mask.to(DEV).to(torch.uint8) if use_mask else None
    y = AG.SasaCore.apply(qd, bbox.to(DEV), md, tuple(S.PC_RANGE), H, p, seed)
    y.backward(gy.to(DEV))
    qc = qkvt.double().requires_grad_(True)
    yc = _sasa_ref(qc, bbox, mask, H, keep, p)
    yc.backward(gy.double())
    assert rel(y, yc) < 2e-5
    assert rel(qd.grad, qc.grad) < 5e-5
    if p > 0:                                            # the mask really drops ~p of the probabilities
        assert abs(1 - keep.float().mean().item() - p) < 0.02


@torch.enable_grad()
@pytest.mark.parametrize('T', [2, 8])
def test_adaptive_mixing_function_vs_oracle_autograd(T):
    from oracle import sparsebev_oracle as O
    g = torch.Generator().manual_seed(T)
    B, Q, G, P, C = 1, 20, 4, 4, 64
    params = S.make_params(31, embed_dims=256, num_frames=T, num_points=P, num_levels=4)
    names = ['mixing.parameter_generator.weight', 'mixing.parameter_generator.bias', 'mixing.out_proj.weight', 'mixing.out_proj.bias']
    x = torch.randn(B, Q, G, T * P, C, generator=g)
    query = torch.randn(B, Q, 256, generator=g)
    gy = torch.randn(B, Q, 256, generator=g)
    dv = [t.to(DEV).requires_grad_(True) for t in [x, query] + [params[n] for n in names]]
    y = AG.AdaptiveMixing.apply(*dv, 128, T == 8)            # T = 8: the recompute variant, T = 2: activations kept
    y.backward(gy.to(DEV))
    pc = {n: params[n].double().requires_grad_(True) for n in names}
    xc, qc = x.double().requires_grad_(True), query.double().requires_grad_(True)
    yc = O.adaptive_mixing(pc, xc, qc)
    yc.backward(gy.double())
    assert rel(y, yc) < 2e-5
    for d, c in zip(dv, [xc, qc] + [pc[n] for n in names]):
        assert rel(d.grad, c.grad) < 1e-4, (d.shape, rel(d.grad, c.grad))


@torch.enable_grad()
def test_adaptive_mixing_fp16_gemms_full_size_without_a_tap_match_the_exact_path():
    """autograd.AdaptiveMixing at config 2's size (900 queries: the grad_W products qualify for sbev_gemm_tn_f16s) with gemm_f16=True and
    NO ParamTap -- the gradients are returned as new tensors -- against the same node on the exact f32-MFMA kernels, and with a gradient
    of tiny magnitude (the scales are measured on the device: nothing may underflow)."""
    g = torch.Generator().manual_seed(77)
    B, Q, G, T, P, C = 1, 900, 4, 8, 4, 64
    params = S.make_params(31, embed_dims=256, num_frames=T, num_points=P, num_levels=4)
    names = ['mixing.parameter_generator.weight', 'mixing.parameter_generator.bias', 'mixing.out_proj.weight', 'mixing.out_proj.bias']
    x = torch.randn(B, Q, G, T * P, C, generator=g).to(DEV)
    query = torch.randn(B, Q, 256, generator=g).to(DEV)
    for mag in (1.0, 1e-9):
        gy = (torch.randn(B, Q, 256, generator=g) * mag).to(DEV)
        grads = []
        for f16 in (False, True):
            dv = [x.clone().requires_grad_(True), query.clone().requires_grad_(True)] + [params[n].to(DEV).requires_grad_(True) for n in names]
            y = AG.AdaptiveMixing.apply(*dv, 128, False, f16)
            y.backward(gy)
            grads.append([y.detach()] + [d.grad for d in dv])
        for k, (a, b) in enumerate(zip(*grads)):
            assert torch.isfinite(b).all()
            if k == 0:
                assert rel(b, a) < 2e-5, (mag, rel(b, a))          # the forward output
                continue
            # gradients: the two forwards differ in the last bits, so a handful of the 37 M ReLU decisions of the mixing core flip and
            # move their (query, group) block through its LayerNorm statistics (measured: 7 of 3600 blocks) -- norm-wise agreement, and
            # all but a small fraction element-wise
            d = (b.double() - a.double()).abs()
            l2 = (d.pow(2).sum() / a.double().pow(2).sum()).sqrt().item()
            off = (d > 2e-5 * a.abs().max().double()).double().mean().item()
            assert l2 < 2e-3 and (k != 1 or off < 1e-2), (mag, k, a.shape, l2, off)      # (element-wise: grad_x only; the others sum over blocks)


@torch.enable_grad()
@pytest.mark.parametrize('T,L,pyr', [(2, 4, 'tiny'), (4, 5, 'tiny5')])
def test_sampling_function_vs_oracle_autograd(T, L, pyr):
    """sample points -> projection / view select -> gather, differentiated: grads wrt the box (centre, dims, yaw), the packed
    offsets | level logits and the feature maps vs the oracle's torch ops under autograd (fp32 on the CPU)."""
    from oracle import sparsebev_oracle as O
    B, Q, G, P = 2, 36, 4, 4
    ih, iw, sizes = S.PYRAMIDS[pyr]
    g = torch.Generator().manual_seed(T * 10 + L)
    bbox, _ = S.make_queries(B, Q, seed=7)
    both = torch.randn(B, Q, G * P * (3 + L), generator=g) * 0.5
    feats = S.make_features(B, T, sizes, seed=8)
    metas = S.make_img_metas(B, T, ih, iw)
    gy = torch.randn(B, Q, G, T * P, 64, generator=g)
    dev_feats = [f.to(DEV).requires_grad_(True) for f in feats]
    pyrd, ctx = FeaturePyramid(dev_feats), DecoderContext(metas, B, torch.device(DEV))
    bd, sd = bbox.to(DEV).requires_grad_(True), both.to(DEV).requires_grad_(True)
    out = AG.Sampling.apply(bd, sd, pyrd, ctx, (T, G, P, L, tuple(S.PC_RANGE)), AG.feature_token(pyrd, dev_feats))
    out.backward(gy.to(DEV))
    # oracle: the same chain in torch (grid_sample sampler = the reference's native path), autograd
    bc, sc = bbox.clone().requires_grad_(True), both.clone().requires_grad_(True)
    fc = [f.clone().requires_grad_(True) for f in feats]
    td = O.time_diff_from_metas(metas, B)
    l2i = torch.from_numpy(np.asarray([m['lidar2img'] for m in metas]).astype(np.float32))
    n_off = G * P * 3
    off = sc[..., :n_off].reshape(B, Q, G * P, 3)
    pts = O.make_sample_points(bc, off, S.PC_RANGE).reshape(B, Q, 1, G, P, 3).expand(B, Q, T, G, P, 3)
    dist = (bc[..., 8:].detach()[:, :, None, :] * td[:, None, :, None])[:, :, :, None, None, :]
    pts = torch.cat([pts[..., 0:2] - dist, pts[..., 2:3]], dim=-1)
    sw = torch.softmax(sc[..., n_off:].reshape(B, Q, G, 1, P, L), -1).expand(B, Q, G, T, P, L)
    fr = O.regroup_features(fc, channel_last=False)
    ref, _ = O.sampling_4d(pts, fr, sw, l2i, ih, iw, O.msmv_sampling_gridsample)
    ref.backward(gy)
    assert rel(out, ref) < 1e-4
    assert rel(bd.grad, bc.grad) < 2e-4 and bd.grad[..., 8:].abs().max() == 0
    assert rel(sd.grad, sc.grad) < 2e-4
    for d, c in zip(dev_feats, fc):
        assert d.grad.shape == c.grad.shape and rel(d.grad, c.grad) < 2e-4


@torch.enable_grad()
def test_refine_bbox_function_vs_torch():
    from oracle import sparsebev_oracle as O
    g = torch.Generator().manual_seed(3)
    B, Q = 2, 50
    bbox = torch.rand(B, Q, 10, generator=g)
    bbox[0, 0, 0], bbox[0, 1, 1] = 0.0, 1.0                   # clamp edges of inverse_sigmoid
    reg = torch.randn(B, Q, 10, generator=g)
    vd = torch.tensor([0.5, 1.0])
    gy = torch.randn(B, Q, 10, generator=g)
    bd, rd = bbox.to(DEV).requires_grad_(True), reg.to(DEV).requires_grad_(True)
    y = AG.RefineBbox.apply(bd, rd, vd.to(DEV))
    y.backward(gy.to(DEV))
    bc, rc = bbox.double().requires_grad_(True), reg.double().requires_grad_(True)
    xyz = torch.sigmoid(rc[..., :3] + O.inverse_sigmoid(bc[..., :3]))
    yc = torch.cat([xyz, rc[..., 3:8], rc[..., 8:] / vd.double()[:, None, None]], -1)
    yc.backward(gy.double())
    assert rel(y, yc) < 1e-5 and rel(rd.grad, rc.grad) < 1e-5
    inner = (bbox[..., :3] > 1e-4) & (bbox[..., :3] < 1 - 1e-4)
    assert ((bd.grad[..., :3].cpu().double() - bc.grad[..., :3]).abs()[inner]).max() < 1e-4 * bc.grad.abs().max()


def build(T, L, seed, num_layers):
    params = S.make_params(seed, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=num_layers, num_levels=L, num_classes=10,
                             code_size=10, pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    return m.to(DEV)


def _g11_run(tag, value_forced=False):
    sample_indices = S.grad_sample_indices
    g = load_golden('g11_train_' + tag)
    B, Q, T, L, n_layers = [int(v) for v in g['cfg']]
    seeds = [int(v) for v in g['seeds']]
    ih, iw, sizes = S.PYRAMIDS[str(g['pyramid'])]
    model = build(T, L, seeds[0], n_layers).train()
    model.decoder.decoder_layer.self_attn.attn_drop = 0.0          # the fixture was recorded with the mmcv dropouts at 0
    model.decoder.decoder_layer.ffn_drop = 0.0
    if value_forced:
        model.decoder.value_forcing = (g['out_bbox'], g['out_feat'])
    feats = [f.to(DEV).requires_grad_(True) for f in S.make_features(B, T, sizes, seed=seeds[2])]
    metas = S.make_img_metas(B, T, ih, iw)
    for b, m in enumerate(metas):
        m['img_timestamp'] = [float(v) for v in g['timestamps'][b]]
    bbox, feat = g['query_bbox'].to(DEV).requires_grad_(True), g['query_feat'].to(DEV).requires_grad_(True)
    cls, box = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
    assert cls.requires_grad and box.requires_grad
    ((cls * g['cot_cls'].to(DEV)).sum() + (box * g['cot_box'].to(DEV)).sum()).backward()
    got = {'query_bbox': bbox.grad, 'query_feat': feat.grad}
    errs = {'out_cls': rel(cls, g['out_cls']), 'out_bbox': rel(box, g['out_bbox']),
            'grad_query_bbox': rel(bbox.grad, g['grad_query_bbox']), 'grad_query_feat': rel(feat.grad, g['grad_query_feat'])}
    named = [(k[len(PREFIX):], p.grad) for k, p in model.named_parameters()] + [('feat%d' % i, f.grad) for i, f in enumerate(feats)]
    assert len(named) == 48 + L
    for name, gr in named:
        assert gr is not None, name
        want = g['g.' + name]
        idx = sample_indices(gr.numel())
        have = gr.reshape(want.shape) if idx is None else gr.reshape(-1)[idx.to(DEV)]
        errs[name] = rel(have, want)
        errs[name + '.norm'] = abs(gr.double().norm().item() - float(g['g.' + name + '.norm'])) / max(float(g['g.' + name + '.norm']), 1e-12)
    return errs, got


@torch.enable_grad()
@pytest.mark.parametrize('tag', ['L2', 'L6'])
def test_g11_train_mode_gradients_match_the_reference_autograd(tag):
    """model.train() forward + backward of the drop-in module (2 layers at B = 2; 6 layers at B = 1; T = 2) against the
    REFERENCE's own modules differentiated by torch autograd: outputs, d/d query_feat, d/d query_bbox, all 48 parameters and
    the feature maps, to 1e-4 relative (max-abs error over max-abs value, per tensor) at two layers.

    ReLU / LayerNorm masks make the gradient a discontinuous function of a layer's input, and fp32 rounding noise in that
    input grows ~5x per random-init layer between ANY two implementations (the fp64 and fp32 CPU oracles differ from each
    other by the same 4e-3 after two layers, tools/exp/debug_g11.py).  The comparison is therefore VALUE-FORCED: layer i+1
    is evaluated at the reference's recorded layer-i outputs (x + (x_ref - x).detach(): values replaced, autograd graph
    intact), which keeps every layer at the reference's operating point while the full multi-layer chain rule -- shared
    weights accumulating over layers, the detach of the refined boxes, the shared feature-gradient buffer -- is exercised.

    Six layers: the value-forced 6-layer gradient is ill-conditioned in fp32 for ANY implementation -- the reference's own
    fixture differs from the fp64 evaluation of the same value-forced chain by 1.2e-3 (worst tensor) / 1e-4 (median)
    (tools/exp/debug_l6_conditioning.py): the backward pass of a random-init decoder amplifies like its forward pass (~5x
    per layer), so a rounding difference entering at layer 6 is three orders of magnitude larger by layer 1.  The HIP path
    adds two sources the CPU reference does not have: hardware-approximated exp / rsqrt / log (1-2 ulp instead of 0.5) in
    LayerNorm and softmax, and a discontinuity INSIDE a layer -- the first-hit camera of a sample point (sampling_4d's argmax
    over the hit mask) is chosen from 3-D points that come out of device expf / sinf / cosf / atan2f, ulps away from the CPU's,
    so about one point in 10^3..10^4 that projects onto an image border picks another camera (the projection
    itself is bit-exact).  Measured: merely re-ordering the partial sums of the attention backward (itself within 1.5e-5 of
    fp64 at every layer's operating point, tools/exp/debug_sasa_l6.py) moved the worst tensor from 2e-3 to 1.5e-2.  Hence the
    loose six-layer bound (median tensor < 1e-3, worst < 5e-2); the strict 1e-4 statement is the two-layer fixture."""
    errs, got = _g11_run(tag, value_forced=True)
    ranked = sorted(errs.items(), key=lambda kv: -kv[1])
    if tag == 'L2':
        assert ranked[0][1] < 1e-4, ranked[:8]
    else:
        vals = sorted(errs.values())
        assert vals[len(vals) // 2] < 1e-3 and ranked[0][1] < 5e-2, ranked[:8]
    assert got['query_bbox'][..., 8:].abs().max() == 0          # velocity is detached (:288) and refine reads reg only


@torch.enable_grad()
@pytest.mark.parametrize('tag,bound', [('L2', 2e-2), ('L6', 2e-1)])
def test_g11_free_running_gradients_stay_close_to_the_reference(tag, bound):
    """The same without value forcing (what a training step really runs): bounded loosely, see above."""
    errs, _ = _g11_run(tag)
    worst = max(errs.items(), key=lambda kv: kv[1])
    assert worst[1] < bound, sorted(errs.items(), key=lambda kv: -kv[1])[:8]


@torch.enable_grad()
def test_feature_gradient_with_partial_layer_loss_and_repeated_backward():
    """The feature-map gradient is handed over by ONE node per decoder call (autograd.FeatureTap) that autograd runs after exactly
    the sampler backwards the current pass reaches (a forward-time counter lost the gradient when the loss touched
    only some layers and went negative on a second backward).  (a) a loss on layer 0's scores only = the 1-layer module's
    gradient; (b) two backward passes under retain_graph accumulate twice the gradient, none is dropped or kept back."""
    B, Q, T, L = 1, 36, 2, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=16)]
    metas = S.make_img_metas(B, T, ih, iw)
    base = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=17)]

    def run(num_layers, loss_of, passes=1):
        model = build(T, L, 15, num_layers).eval()
        fs = [f.clone().requires_grad_(True) for f in base]
        cls, box = model(bbox, feat, list(fs), None, copy.deepcopy(metas))
        loss = loss_of(cls, box)
        for i in range(passes):
            loss.backward(retain_graph=i + 1 < passes)
        return [f.grad for f in fs]

    # (a) only layer 0 of a 2-layer decoder is in the loss: layer 1's Sampling node never runs its backward
    part = run(2, lambda c, b: c[0].sum() + b[0].pow(2).sum())
    one = run(1, lambda c, b: c[0].sum() + b[0].pow(2).sum())
    for gp, go in zip(part, one):
        assert gp is not None and torch.isfinite(gp).all() and gp.abs().max() > 0
        assert (gp - go).abs().max() <= 1e-5 * go.abs().max()          # (atomics: summation order differs run to run)
    # (b) the same graph differentiated twice
    once = run(2, lambda c, b: c.sum() + b.sum())
    twice = run(2, lambda c, b: c.sum() + b.sum(), passes=2)
    for g1, g2 in zip(once, twice):
        assert g2 is not None and (g2 - 2 * g1).abs().max() <= 1e-5 * g1.abs().max()
    # (c) a loss that reaches no Sampling node at all leaves the features without gradient (and nothing allocated behind)
    none = run(2, lambda c, b: (c * 0).sum().detach().requires_grad_(True))
    assert all(g is None for g in none)


@torch.enable_grad()
def test_shared_parameter_gradients_collected_per_call_equal_autograd_accumulation():
    """autograd.Tap / ParamTap (round 3): the parameters the layers share collect their gradient in one buffer per call, added to in the
    kernels' epilogues, and reach ``.grad`` through ONE node -- the same numbers as autograd's own accumulation (SBEV tap off) to
    summation-order rounding; every parameter gets a gradient; a loss on layer 0 only equals the 1-layer module's gradients; two backward
    passes under retain_graph give twice the gradient."""
    B, Q, T, L = 1, 36, 2, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=26)]
    metas = S.make_img_metas(B, T, ih, iw)
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=27)]

    def run(num_layers, tap, loss_of=lambda c, b: c.sum() + b.pow(2).sum(), passes=1):
        model = build(T, L, 25, num_layers).eval()
        model.decoder.tap_param_grads = tap
        cls, box = model(bbox, feat.clone().requires_grad_(True), list(feats), None, copy.deepcopy(metas))
        loss = loss_of(cls, box)
        for i in range(passes):
            loss.backward(retain_graph=i + 1 < passes)
        return {n: p.grad for n, p in model.named_parameters()}

    a, b = run(3, True), run(3, False)
    assert all(g is not None for g in a.values()) and set(a) == set(b)
    for n in a:
        assert (a[n] - b[n]).abs().max() <= 2e-5 * max(b[n].abs().max().item(), 1e-3), n
    first = lambda c, bx: c[0].sum() + bx[0].pow(2).sum()
    part, one = run(2, True, first), run(1, True, first)
    for n in part:
        if one[n] is None:
            assert part[n] is None or float(part[n].abs().max()) == 0.0, n
        else:
            assert (part[n] - one[n]).abs().max() <= 2e-5 * max(one[n].abs().max().item(), 1e-3), n
    twice = run(3, True, passes=2)
    for n in a:
        assert (twice[n] - 2 * a[n]).abs().max() <= 2e-5 * max(a[n].abs().max().item(), 1e-3), n
    # more layers than one grouped launch reduces (8 segments): the second chunk accumulates in a later launch
    a9, b9 = run(9, True), run(9, False)
    for n in a9:
        assert (a9[n] - b9[n]).abs().max() <= 5e-5 * max(b9[n].abs().max().item(), 1e-3), n


@torch.enable_grad()
def test_grouped_branch_launches_equal_one_launch_per_linear():
    """Round 6 (autograd.LinearPair): the classification and regression heads level by level in ONE grouped launch each way -- forward
    outputs bit-identical to one Linear node per head (the grouped kernel runs the same tile code), parameter and query gradients equal to
    summation-order rounding; with and without the parameter tap."""
    import os
    B, Q, T, L = 1, 64, 2, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=36)]
    metas = S.make_img_metas(B, T, ih, iw)
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=37)]

    def run(pairs, tap):
        prev = os.environ.pop('SBEV_NO_TRAIN_PAIRS', None)
        if not pairs:
            os.environ['SBEV_NO_TRAIN_PAIRS'] = '1'
        try:
            model = build(T, L, 35, 2).eval()
            model.decoder.tap_param_grads = tap
            q = feat.clone().requires_grad_(True)
            cls, box = model(bbox, q, list(feats), None, copy.deepcopy(metas))
            (cls.sum() + box.pow(2).sum()).backward()
            return cls.detach(), box.detach(), q.grad, {n: p.grad for n, p in model.named_parameters()}
        finally:
            os.environ.pop('SBEV_NO_TRAIN_PAIRS', None)
            if prev is not None:
                os.environ['SBEV_NO_TRAIN_PAIRS'] = prev

    for tap in (True, False):
        a, b = run(True, tap), run(False, tap)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert (a[2] - b[2]).abs().max() <= 2e-5 * max(b[2].abs().max().item(), 1e-3)
        assert set(a[3]) == set(b[3]) and all(g is not None for g in a[3].values())
        for n in a[3]:
            assert (a[3][n] - b[3][n]).abs().max() <= 2e-5 * max(b[3][n].abs().max().item(), 1e-3), n


@torch.enable_grad()
@pytest.mark.parametrize('feat_grad', [False, True])
def test_captured_training_step_replays_with_new_data_and_equals_the_eager_step(feat_grad):
    """sparsebev_amd.train_graph.CapturedTrainStep: forward + backward of a 3-layer decoder captured as ONE hipGraph on batch A, then
    replayed on batch B (new features / queries copied into the static tensors, new camera matrices and time stamps through
    ``replay(img_metas)``): loss, outputs and every gradient equal an eager step on batch B (1e-5 relative: the feature-gradient
    scatter and the split-K reductions add in another order from run to run), and a second replay of B reproduces the first."""
    from sparsebev_amd.train_graph import CapturedTrainStep
    B, Q, T, layers = 1, 100, 2, 3
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, len(sizes), 77, layers).train()
    model.decoder.decoder_layer.self_attn.attn_drop = 0.0
    model.decoder.decoder_layer.ffn_drop = 0.0
    cot = [torch.randn(layers, B, Q, 10, generator=torch.Generator().manual_seed(5 + i)).to(DEV) for i in range(2)]
    loss_fn = lambda cls, box: (cls * cot[0]).sum() + (box * cot[1]).sum()

    def batch(seed):
        feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=seed)]
        bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=seed + 1)]
        metas = S.make_img_metas(B, T, ih, iw)
        for m in metas:
            m['img_timestamp'] = [t_ - 0.013 * seed * (i // 6) for i, t_ in enumerate(m['img_timestamp'])]
            m['lidar2img'] = [np.asarray(a, np.float32) * (1.0 + 1e-3 * seed) for a in m['lidar2img']]
        return feats, bbox, feat, metas

    fa, ba, qa, ma = batch(1)
    sf = [f.clone().requires_grad_(feat_grad) for f in fa]
    sb, sq = ba.clone(), qa.clone().requires_grad_(True)
    step = CapturedTrainStep(model, sb, sq, sf, ma, loss_fn)
    assert len(step.grads) == 48
    fb, bb, qb, mb = batch(2)
    with torch.no_grad():
        for d, s_ in zip(sf, fb):
            d.copy_(s_)
        sb.copy_(bb)
        sq.copy_(qb)
    loss, cls, box = [t.clone() for t in step.replay(mb)]
    g1 = {n: t.clone() for n, t in step.grads.items()}
    gq = step.input_grads['query_feat'].clone()
    gf = [t.clone() for t in step.input_grads['mlvl_feats']] if feat_grad else []
    loss2 = step.replay()[0].clone()
    assert rel(loss2, loss) < 1e-6 and all(rel(step.grads[n], g1[n]) < 1e-5 for n in g1)
    del step
    # eager on batch B, fresh leaves
    for p in model.parameters():
        p.grad = None
    ef = [f.clone().requires_grad_(feat_grad) for f in fb]
    eq = qb.clone().requires_grad_(True)
    ecls, ebox = model(bb, eq, list(ef), None, copy.deepcopy(mb))
    eloss = loss_fn(ecls, ebox)
    eloss.backward()
    assert rel(cls, ecls) < 1e-6 and rel(box, ebox) < 1e-6 and rel(loss, eloss) < 1e-6
    assert rel(gq, eq.grad) < 1e-5
    for n, p in model.decoder.named_parameters():
        assert rel(g1[n], p.grad) < 1e-5, n
    for a_, b_ in zip(gf, ef):
        assert rel(a_, b_.grad) < 1e-5
    # and the two batches really differ
    assert g1['decoder_layer.ffn.layers.1.weight'].abs().max() > 0 and rel(qa, qb) > 0.1


@torch.enable_grad()
def test_captured_training_step_with_dropout_draws_new_masks_per_replay_and_equals_eager_for_the_same_seeds():
    """train() mode, mmcv's dropouts (0.1) on: the captured step hashes host seed (frozen at capture) + a device word that ``replay``
    re-draws -- two replays differ, a replay with the same word repeats bit for bit, and an eager step with the same host seeds (same
    CPU generator state) and the same device word gives the same loss and gradients."""
    from sparsebev_amd.train_graph import CapturedTrainStep
    B, Q, T, layers = 1, 100, 2, 2
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, len(sizes), 79, layers).train()
    assert model.decoder.decoder_layer.self_attn.attn_drop > 0 and model.decoder.decoder_layer.ffn_drop > 0
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=3)]
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=4)]
    metas = S.make_img_metas(B, T, ih, iw)
    loss_fn = lambda cls, box: cls.sum() + box.sum()
    torch.manual_seed(7)
    sq = feat.clone().requires_grad_(True)
    step = CapturedTrainStep(model, bbox, sq, feats, metas, loss_fn, warmup=1)
    assert step.dropout and step.seed_dev is not None
    l1 = step.replay()[0].clone()
    w1 = step.seed_dev.clone()
    l2 = step.replay()[0].clone()
    assert not torch.equal(step.seed_dev, w1) and abs(l1.item() - l2.item()) > 1e-6 * abs(l1.item())
    step.seed_dev.copy_(w1)
    l1b = step.replay(new_masks=False)[0].clone()
    assert torch.equal(l1, l1b)
    g1 = {n: t.clone() for n, t in step.grads.items()}
    del step
    # eager: the capture was the second step after manual_seed(7) (one warm-up step = `layers` draws of a host seed before it)
    torch.manual_seed(7)
    for _ in range(layers):
        torch.randint(0, 2 ** 62, (1,))
    for p in model.parameters():
        p.grad = None
    eq = feat.clone().requires_grad_(True)
    with AG.device_seed(w1):
        ecls, ebox = model(bbox, eq, list(feats), None, copy.deepcopy(metas))
        eloss = loss_fn(ecls, ebox)
        eloss.backward()
    assert rel(l1, eloss) < 1e-6
    for n, p in model.decoder.named_parameters():
        assert rel(g1[n], p.grad) < 1e-5, n


@torch.enable_grad()
def test_tap_with_one_linear_at_two_row_counts_and_an_aborted_pass():
    """a shared Linear called with DIFFERENT row counts inside one tapped call -- the bias gradient's recorded segments are
    its own (a node that adds its bias directly still records its grad_y for the weight: borrowing the weight's list counted that
    bias twice) -- and a backward pass that raises before ParamTap runs must not leak its partial sums into the retry."""
    g = torch.Generator().manual_seed(77)
    w = (torch.randn(256, 256, generator=g) / 16).to(DEV).requires_grad_(True)
    b = torch.randn(256, generator=g).to(DEV).requires_grad_(True)
    gam = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV).requires_grad_(True)
    bet = (0.1 * torch.randn(256, generator=g)).to(DEV).requires_grad_(True)
    xs = [torch.randn(m, 256, generator=g).to(DEV) for m in (36, 20, 36, 36, 20)]

    def loss_of(lin, ln):
        ys = [ln(lin(x, w, b), gam, bet) for x in xs]
        return sum((y * (i + 1)).pow(2).sum() for i, y in enumerate(ys))

    ref = loss_of(lambda x, w_, b_: F.relu(F.linear(x, w_, b_)), lambda y, g_, b_: F.layer_norm(y, (256,), g_, b_, 1e-5))
    want = torch.autograd.grad(ref, [w, b, gam, bet])
    tap = AG.Tap([w, b, gam, bet])
    out = loss_of(lambda x, w_, b_: AG.linear(x, w_, b_, relu=True, tap=tap), lambda y, g_, b_: AG.layer_norm(y, g_, b_, tap=tap))
    assert abs(out.item() - ref.item()) <= 1e-4 * abs(ref.item())

    w.grad = b.grad = gam.grad = bet.grad = None
    out.backward(retain_graph=True)
    for p, r, name in zip((w, b, gam, bet), want, ('weight', 'bias', 'gamma', 'beta')):
        assert (p.grad - r).abs().max() <= 2e-5 * r.abs().max().item(), name

    class Boom(torch.autograd.Function):                      # sits UPSTREAM of the first Linear: its backward runs after every tapped
        armed = True                                          # node has recorded / accumulated, before ParamTap

        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, gr):
            if Boom.armed:
                raise RuntimeError('boom')
            return gr
    leaf = xs[0].clone().requires_grad_(True)
    xs[0] = Boom.apply(leaf)
    tap = AG.Tap([w, b, gam, bet])
    out = loss_of(lambda x, w_, b_: AG.linear(x, w_, b_, relu=True, tap=tap), lambda y, g_, b_: AG.layer_norm(y, g_, b_, tap=tap))
    w.grad = b.grad = gam.grad = bet.grad = None
    with pytest.raises(RuntimeError, match='boom'):
        out.backward(retain_graph=True)
    Boom.armed = False
    w.grad = b.grad = gam.grad = bet.grad = None
    out.backward()
    for p, r, name in zip((w, b, gam, bet), want, ('weight', 'bias', 'gamma', 'beta')):
        assert (p.grad - r).abs().max() <= 2e-5 * r.abs().max().item(), name


@torch.enable_grad()
@pytest.mark.parametrize('reentrant', [True, False])
def test_tap_survives_nested_backward_passes_of_reentrant_checkpointing(reentrant):
    """``torch.utils.checkpoint(use_reentrant=True)`` runs an INNER backward pass (its own graph-task id) inside the outer
    one.  Round 4's Tap discarded every pending partial sum whenever the task id changed, so tapped nodes outside the checkpoints
    lost their weight / bias / LayerNorm contributions -- parameter gradients silently too small.  The same shared Linear +
    LayerNorm at two row counts, some calls inside checkpoints and some outside: gradients equal plain autograd's."""
    from torch.utils.checkpoint import checkpoint
    g = torch.Generator().manual_seed(78)
    w = (torch.randn(256, 256, generator=g) / 16).to(DEV).requires_grad_(True)
    b = torch.randn(256, generator=g).to(DEV).requires_grad_(True)
    gam = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV).requires_grad_(True)
    bet = (0.1 * torch.randn(256, generator=g)).to(DEV).requires_grad_(True)
    xs = [torch.randn(m, 256, generator=g).to(DEV).requires_grad_(True) for m in (36, 20, 36, 36, 20, 36)]
    inside = (False, True, False, True, True, False)

    def ref_block(x):
        return F.layer_norm(F.relu(F.linear(x, w, b)), (256,), gam, bet, 1e-5)
    ref = sum((ref_block(x) * (i + 1)).pow(2).sum() for i, x in enumerate(xs))
    want = torch.autograd.grad(ref, [w, b, gam, bet] + xs)

    tap = AG.Tap([w, b, gam, bet])

    def block(x):
        return AG.layer_norm(AG.linear(x, w, b, relu=True, tap=tap), gam, bet, tap=tap)
    ys = [checkpoint(block, x, use_reentrant=reentrant) if ck else block(x) for x, ck in zip(xs, inside)]
    out = sum((y * (i + 1)).pow(2).sum() for i, y in enumerate(ys))
    assert abs(out.item() - ref.item()) <= 1e-4 * abs(ref.item())
    for t in [w, b, gam, bet] + xs:
        t.grad = None
    out.backward()
    for p, r, name in zip([w, b, gam, bet] + xs, want, ['weight', 'bias', 'gamma', 'beta'] + ['x%d' % i for i in range(len(xs))]):
        assert p.grad is not None, name
        assert (p.grad - r).abs().max() <= 2e-5 * r.abs().max().item(), (name, (p.grad - r).abs().max().item(), r.abs().max().item())


@torch.enable_grad()
def test_eval_mode_with_grad_is_differentiable_and_matches_the_inference_runtime():
    """Grad enabled + something requires grad -> the module is differentiable like the reference's (no silent detached
    outputs); its forward values equal the fused inference runtime's to rounding; under no_grad the runtime runs."""
    B, Q, T, L = 1, 36, 2, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, L, 5, 2).eval()
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=6)]
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=7)]
    metas = S.make_img_metas(B, T, ih, iw)
    cls, box = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
    assert cls.grad_fn is not None
    with torch.no_grad():
        cls_i, box_i = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
    assert cls_i.grad_fn is None
    assert (cls - cls_i).abs().max() < 1e-4 and (box - box_i).abs().max() < 1e-4
    (cls.sum() + box.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@torch.enable_grad()
def test_train_mode_dropout_is_active_seeded_and_unbiased():
    B, Q, T, L = 1, 64, 2, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, L, 9, 1).train()
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=10)]
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=11)]
    metas = S.make_img_metas(B, T, ih, iw)
    torch.manual_seed(1)
    a = model(bbox, feat, list(feats), None, copy.deepcopy(metas))[0]
    torch.manual_seed(1)
    b = model(bbox, feat, list(feats), None, copy.deepcopy(metas))[0]
    c = model(bbox, feat, list(feats), None, copy.deepcopy(metas))[0]
    assert torch.equal(a, b)                                   # same torch seed -> same masks
    assert not torch.equal(a, c)                               # dropout is on in train()
    model.eval()
    d = model(bbox, feat, list(feats), None, copy.deepcopy(metas))[0]
    assert (a - d).abs().max() > 1e-4 and (a - d).abs().mean() < 0.5
    # the elementwise dropout kernel: keep rate and scaling
    x = torch.ones(1 << 20, device=DEV)
    y = AG.dropout(x, 0.1, 42)
    assert abs((y == 0).float().mean().item() - 0.1) < 5e-3 and abs(y.mean().item() - 1.0) < 1e-2
    assert torch.equal(y, AG.dropout(x, 0.1, 42)) and not torch.equal(y, AG.dropout(x, 0.1, 43))


@torch.enable_grad()
def test_full_size_c2_one_layer_backward_vs_oracle_autograd():
    """BASELINE config 2 at full size (r50 704x256 pyramid, 900 queries, T = 8), one layer, forward + backward against the
    oracle's torch autograd on the CPU: the backward kernels at the workload's real shapes (M = 900 rows: ragged 128-row GEMM
    tiles, K = 900 reductions, the 64-way split-K of the generator's input gradient, 3 600 mixing items, 115 200 sample points).
    The gradient is discontinuous wherever the forward takes a decision -- the ReLU of the FFN's 460 800 hidden activations, the
    LayerNorm+ReLU masks of adaptive mixing, the first-hit camera of the 115 200 sample points -- and the device's values are
    ulps away from the CPU's, so at this size a handful of decisions flip: ONE flipped FFN activation moves its row of the
    FFN weight gradient by ~1/sqrt(900) of the maximum (measured: 1.2e-2 on ffn.layers.0.0.weight with everything else at
    1e-6).  Hence: the tensors with no decision between them and the loss (norm3, both branches) agree to 1e-4; every tensor
    agrees NORM-wise to 2e-2 (measured 4.8e-3 on the sampling offsets, which see every flipped camera); max-abs errors are bounded
    at 5e-2 with the median tensor at 1e-2.  The strict per-tensor 1e-4 statement is the two-layer fixture G11 (small enough to see no flip)."""
    from oracle import sparsebev_oracle as O
    B, Q, T, L = 1, 900, 8, 4
    ih, iw, sizes = S.PYRAMIDS['r50_704x256']
    params = S.make_params(140, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    model = build(T, L, 140, 1).eval()
    feats = S.make_features(B, T, sizes, seed=141)
    bbox, feat = S.make_queries(B, Q, seed=142)
    metas = S.make_img_metas(B, T, ih, iw)
    g = torch.Generator().manual_seed(143)
    cc, cb = torch.randn(1, B, Q, 10, generator=g), torch.randn(1, B, Q, 10, generator=g)
    fd = feat.to(DEV).requires_grad_(True)
    cls, box = model(bbox.to(DEV), fd, [f.to(DEV) for f in feats], None, copy.deepcopy(metas))
    ((cls * cc.to(DEV)).sum() + (box * cb.to(DEV)).sum()).backward()
    fo = feat.clone().requires_grad_(True)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    c2, b2, _ = O.decoder(po, bbox, fo, feats, metas, S.PC_RANGE, num_layers=1)
    ((c2 * cc).sum() + (b2 * cb).sum()).backward()
    assert rel(cls, c2) < 1e-4 and rel(box, b2) < 1e-4
    def rel_l2(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

    pairs = [('query_feat', fd.grad, fo.grad)] + [(k[len(PREFIX):], p.grad, po[k[len(PREFIX):]].grad) for k, p in model.named_parameters()]
    l2 = {k: rel_l2(a, b) for k, a, b in pairs}
    mx = {k: rel(a, b) for k, a, b in pairs}
    branches = {k: v for k, v in mx.items() if 'cls_branch' in k or 'reg_branch' in k or 'norm3' in k}
    report = sorted(((k, l2[k], mx[k]) for k in l2), key=lambda t: -t[1])[:8]
    assert max(branches.values()) < 1e-4, report                 # no ReLU / camera decision between them and the loss that could flip
    assert max(l2.values()) < 2e-2, report                       # norm-wise: a flipped row / point is one of many (measured: 4.8e-3 worst)
    vals = sorted(mx.values())
    assert vals[len(vals) // 2] < 1e-2 and vals[-1] < 5e-2, report


@torch.enable_grad()
def test_training_refuses_inference_only_feature_formats_and_supports_the_dn_mask():
    """bf16 storage and the online frame ring are inference formats: the differentiable path must refuse them loudly instead of
    silently detaching; the query-denoising attention mask (models/sparsebev_transformer.py:224-225) trains."""
    from sparsebev_amd.cache import FrameFeatureCache
    B, Q, T, L = 1, 36, 2, 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    model = build(T, L, 21, 1).train()
    model.decoder.decoder_layer.self_attn.attn_drop = 0.0
    model.decoder.decoder_layer.ffn_drop = 0.0
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=22)]
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=23)]
    metas = S.make_img_metas(B, T, ih, iw)
    with pytest.raises(NotImplementedError):
        model(bbox, feat, [f.to(torch.bfloat16).permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats], None, copy.deepcopy(metas))
    ring = FrameFeatureCache(T)
    for t in range(T):
        ring.push([f[:, t * 6:(t + 1) * 6].contiguous() for f in feats])
    with pytest.raises(NotImplementedError):
        model(bbox, feat, ring.pyramid(), None, copy.deepcopy(metas))
    mask = torch.zeros(Q, Q, dtype=torch.bool, device=DEV)
    mask[:10, 10:] = True
    mask[10:, :10] = True
    fd = feat.clone().requires_grad_(True)
    cls, box = model(bbox, fd, list(feats), mask, copy.deepcopy(metas))
    (cls.sum() + box.sum()).backward()
    assert torch.isfinite(fd.grad).all() and fd.grad.abs().max() > 0
    # against the oracle's autograd with the same mask
    from oracle import sparsebev_oracle as O
    params = S.make_params(21, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    fo = feat.cpu().clone().requires_grad_(True)
    c2, b2, _ = O.decoder(params, bbox.cpu(), fo, [f.cpu() for f in feats], metas, S.PC_RANGE, num_layers=1, pre_attn_mask=mask.cpu())
    (c2.sum() + b2.sum()).backward()
    assert rel(cls, c2) < 1e-4 and rel(fd.grad, fo.grad) < 1e-3


@torch.enable_grad()
def test_a_few_optimizer_steps_follow_the_oracle_trajectory_and_reduce_the_loss():
    """The module TRAINS: 12 Adam steps on a synthetic regression target (queries and decoder parameters learn; dropouts off for
    determinism) through the HIP forward + backward, beside the same 12 steps of the CPU oracle under torch autograd from the same
    initial state.  The loss must fall, and the two loss trajectories must agree closely early on (they drift apart slowly, as
    any two fp32 implementations of a non-smooth network do)."""
    from oracle import sparsebev_oracle as O
    B, Q, T, L, n_layers, steps = 1, 36, 2, 4, 2, 12
    ih, iw, sizes = S.PYRAMIDS['tiny']
    params0 = S.make_params(301, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    feats = S.make_features(B, T, sizes, seed=302)
    bbox0, feat0 = S.make_queries(B, Q, seed=303)
    metas = S.make_img_metas(B, T, ih, iw)
    g = torch.Generator().manual_seed(304)
    tgt_cls, tgt_box = torch.randn(n_layers, B, Q, 10, generator=g) * 0.5, torch.rand(n_layers, B, Q, 10, generator=g)

    def loss_of(cls, box, tc, tb):
        return ((cls - tc) ** 2).mean() + ((box - tb) ** 2).mean()

    # HIP path
    model = build(T, L, 301, n_layers).train()
    model.decoder.decoder_layer.self_attn.attn_drop = 0.0
    model.decoder.decoder_layer.ffn_drop = 0.0
    qf = feat0.to(DEV).requires_grad_(True)
    dev_feats = [f.to(DEV) for f in feats]
    opt = torch.optim.Adam(list(model.parameters()) + [qf], lr=2e-4)
    hip_losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        cls, box = model(bbox0.to(DEV), qf, list(dev_feats), None, copy.deepcopy(metas))
        loss = loss_of(cls, box, tgt_cls.to(DEV), tgt_box.to(DEV))
        loss.backward()
        opt.step()
        hip_losses.append(loss.item())
    # oracle path, same recipe on the CPU
    po = {k: v.clone().requires_grad_(True) for k, v in params0.items()}
    qo = feat0.clone().requires_grad_(True)
    opt_o = torch.optim.Adam(list(po.values()) + [qo], lr=2e-4)
    ref_losses = []
    for _ in range(steps):
        opt_o.zero_grad(set_to_none=True)
        c2, b2, _ = O.decoder(po, bbox0, qo, feats, metas, S.PC_RANGE, num_layers=n_layers)
        loss = loss_of(c2, b2, tgt_cls, tgt_box)
        loss.backward()
        opt_o.step()
        ref_losses.append(loss.item())
    assert hip_losses[-1] < 0.9 * hip_losses[0], hip_losses
    assert abs(hip_losses[0] - ref_losses[0]) < 1e-4 * ref_losses[0]
    for a, b in zip(hip_losses[:4], ref_losses[:4]):
        assert abs(a - b) < 2e-3 * b, (hip_losses, ref_losses)
    assert abs(hip_losses[-1] - ref_losses[-1]) < 5e-2 * ref_losses[-1], (hip_losses, ref_losses)
