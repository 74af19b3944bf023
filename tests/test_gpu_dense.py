"""GPU parity of the dense kernels (csrc/gemm.hip ...) against fp64 math on the CPU."""
import pytest
import torch

from sparsebev_amd import dense

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def ref_linear(x, w, b, relu=False, res=None, ln=None):
    y = x.double() @ w.double().t() + (b.double() if b is not None else 0)
    if relu:
        y = y.clamp(min=0)
    if res is not None:
        y = y + res.double()
    if ln is not None:
        y = torch.nn.functional.layer_norm(y, [y.shape[-1]], ln[0].double(), ln[1].double())
    return y


@pytest.mark.parametrize('M,N,K', [(900, 32768, 256), (900, 256, 256), (900, 768, 256), (900, 112, 256), (900, 512, 256),
                                   (900, 256, 512), (900, 10, 256), (37, 72, 100), (1, 8, 4), (3600, 256, 256), (129, 130, 36)])
@pytest.mark.parametrize('relu,use_res', [(False, False), (True, True)])
def test_linear_shapes(M, N, K, relu, use_res):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if use_res else None
    y = dense.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=relu, residual=res.to(DEV) if use_res else None)
    ref = ref_linear(x, w, b, relu, res)
    assert y.shape == (M, N)
    assert (y.cpu().double() - ref).abs().max() < 2e-5


def test_out_proj_splitk_fused_epilogue():
    # AdaptiveMixing.out_proj: K = 32768, N = 256, + query residual, + LayerNorm (norm2)
    g = torch.Generator().manual_seed(0)
    M, N, K = 900, 256, 32768
    x = torch.randn(M, K, generator=g).clamp(min=0)
    w = (2 * torch.rand(N, K, generator=g) - 1) / K ** 0.5
    b, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    lw, lb = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    y = dense.linear(x.to(DEV), w.to(DEV), b.to(DEV), residual=res.to(DEV), ln=(lw.to(DEV), lb.to(DEV)))
    ref = ref_linear(x, w, b, False, res, (lw, lb))
    assert (y.cpu().double() - ref).abs().max() < 2e-5
    y2 = dense.linear(x.to(DEV), w.to(DEV), b.to(DEV))
    assert (y2.cpu().double() - ref_linear(x, w, b)).abs().max() < 2e-5


@pytest.mark.parametrize('M', [1, 47, 48, 49, 144, 900, 3200, 3600])      # 3200 / 3600 = B*Q of BASELINE configs c3 / c4
@pytest.mark.parametrize('N,K', [(256, 32768), (128, 4096), (384, 2048 + 32)])
def test_splitk_register_tiled_kernel_ragged(M, N, K):
    """N % 128 == 0, K % 32 == 0 goes to gemm_nt_f32_regtile_kernel (48-row x 128-column wave tasks x K splits):
    row remainders, uneven split boundaries (K/32 not divisible by the split count), with and without the fused LN."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    lnw, lnb = torch.randn(N, generator=g), torch.randn(N, generator=g)
    y = dense.linear(x.to(DEV), w.to(DEV), b.to(DEV), residual=res.to(DEV))
    assert (y.cpu().double() - ref_linear(x, w, b, False, res)).abs().max() < 3e-5
    y = dense.linear(x.to(DEV), w.to(DEV), b.to(DEV), residual=res.to(DEV), ln=(lnw.to(DEV), lnb.to(DEV)))
    assert (y.cpu().double() - ref_linear(x, w, b, False, res, (lnw, lnb))).abs().max() < 1e-4


def test_layer_norm_and_relu():
    g = torch.Generator().manual_seed(1)
    x = 3 * torch.randn(2, 450, 256, generator=g) + 1
    lw, lb = torch.randn(256, generator=g), torch.randn(256, generator=g)
    y = dense.layer_norm(x.to(DEV), lw.to(DEV), lb.to(DEV))
    ref = torch.nn.functional.layer_norm(x.double(), [256], lw.double(), lb.double())
    assert (y.cpu().double() - ref).abs().max() < 1e-5
    y = dense.layer_norm(x.to(DEV), lw.to(DEV), lb.to(DEV), relu=True)
    assert (y.cpu().double() - ref.clamp(min=0)).abs().max() < 1e-5


def test_linear_rejects_cpu_and_misaligned():
    with pytest.raises(RuntimeError):
        dense.linear(torch.zeros(4, 8), torch.zeros(4, 8), None)
    with pytest.raises(RuntimeError):
        dense.linear(torch.zeros(4, 3, device=DEV), torch.zeros(4, 3, device=DEV), None)


@pytest.mark.parametrize('Pin,BQ', [(4, 37), (8, 37), (12, 37), (20, 37), (32, 37), (60, 37), (100, 37), (116, 5), (120, 37), (32, 3200), (32, 3600), (120, 1600)])
def test_adaptive_mixing_core_vs_fp64(Pin, BQ):
    import ctypes
    from sparsebev_amd import _lib
    g = torch.Generator().manual_seed(Pin)
    G, C, Pout = 4, 64, 128
    x = torch.randn(BQ, G, Pin, C, generator=g)
    prm = torch.randn(BQ, G, C * C + Pout * Pin, generator=g) * 0.3
    M = prm[..., : C * C].reshape(BQ, G, C, C).double()
    S = prm[..., C * C:].reshape(BQ, G, Pout, Pin).double()
    y = torch.relu(torch.nn.functional.layer_norm(x.double() @ M, [Pin, C]))
    ref = torch.relu(torch.nn.functional.layer_norm(S @ y, [Pout, C]))
    out = torch.empty(BQ, G, Pout, C, device=DEV)
    xd, pd = x.to(DEV), prm.to(DEV)
    st = _lib.load().sbev_adaptive_mixing_f32(ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(pd.data_ptr()),
                                              ctypes.c_void_p(out.data_ptr()), BQ, G, Pin, C, Pout, 1e-5,
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    assert (out.cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize('shape', [(3, 256, 8, 22), (2, 256, 10, 25), (5, 64, 64, 176), (1, 100, 7, 9)])
def test_nchw_to_nhwc(shape):
    n, c, h, w = shape
    x = torch.randn(1, n, c, h, w, device=DEV)
    y = dense.to_channels_last(x)
    assert y.shape == (1, n, h, w, c)
    assert torch.equal(y, x.permute(0, 1, 3, 4, 2).contiguous())


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(3, 256, 8, 22), (2, 256, 10, 25), (5, 64, 64, 176), (1, 100, 7, 9), (2, 256, 64, 176), (1, 136, 4, 8), (2, 8, 1, 4)])
def test_nchw_to_nhwc_two_byte_channels(shape, dtype):
    """bf16 / fp16 STORAGE through the 2-byte relayout (sbev_nchw_to_nhwc_b16: channel pairs interleaved in registers, a 32-bit transpose
    through LDS): pure byte movement -- every bit pattern must arrive, NaN payloads and negative zeros included; vector path
    (channels % 8 == 0, hw % 4 == 0) and the scalar one"""
    n, c, h, w = shape
    bits = torch.randint(-32768, 32767, (1, n, c, h, w), device=DEV, dtype=torch.int16)
    x = bits.view(dtype)
    y = dense.to_channels_last(x)
    assert y.shape == (1, n, h, w, c) and y.dtype == dtype and y.is_contiguous()
    assert torch.equal(y.view(torch.int16), bits.permute(0, 1, 3, 4, 2).contiguous())


def test_position_encoder_first_layer():
    g = torch.Generator().manual_seed(3)
    bbox = torch.rand(2, 450, 10, generator=g)
    w, b = torch.randn(256, 3, generator=g), torch.randn(256, generator=g)
    lw, lb = torch.randn(256, generator=g), torch.randn(256, generator=g)
    y = dense.linear_ln_relu(bbox.to(DEV), w.to(DEV), b.to(DEV), lw.to(DEV), lb.to(DEV))
    ref = torch.relu(torch.nn.functional.layer_norm(bbox[..., :3].double() @ w.double().t() + b.double(), [256], lw.double(), lb.double()))
    assert (y.cpu().double() - ref).abs().max() < 1e-5


@pytest.mark.parametrize('Q,use_mask', [(900, False), (100, True), (37, True)])
def test_sasa_core_vs_fp64(Q, use_mask):
    import math
    from sparsebev_amd import synthetic as S
    g = torch.Generator().manual_seed(Q)
    B, H, D = 2, 8, 256
    x = torch.randn(B, Q, D, generator=g)
    bbox = torch.rand(B, Q, 10, generator=g)
    in_w, in_b = torch.randn(3 * D, D, generator=g) / 16, 0.1 * torch.randn(3 * D, generator=g)
    out_w, out_b = torch.randn(D, D, generator=g) / 16, 0.1 * torch.randn(D, generator=g)
    tau_w, tau_b = 0.02 * torch.randn(H, D, generator=g), 2 * torch.rand(H, generator=g)
    mask = None
    if use_mask:
        mask = torch.rand(Q, Q, generator=g) < 0.3
        mask.fill_diagonal_(False)
    d = lambda t: t.to(DEV) if t is not None else None
    y = dense.scale_adaptive_self_attention(d(bbox), d(x), S.PC_RANGE, H, d(in_w), d(in_b), d(out_w), d(out_b), d(tau_w), d(tau_b), d(mask))
    # fp64 reference
    xd = x.double()
    cx = bbox[..., 0].double() * 102.4 - 51.2
    cy = bbox[..., 1].double() * 102.4 - 51.2
    xy = torch.stack([cx, cy], -1)
    dist = -(xy[:, :, None] - xy[:, None]).norm(dim=-1)
    tau = xd @ tau_w.double().t() + tau_b.double()
    bias = dist[:, None] * tau.permute(0, 2, 1)[..., None]
    if mask is not None:
        bias = bias.masked_fill(mask[None, None], float('-inf'))
    qkv = xd @ in_w.double().t() + in_b.double()
    q, k, v = (t.reshape(B, Q, H, 32).permute(0, 2, 1, 3) for t in qkv.chunk(3, -1))
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(32) + bias, -1) @ v
    ref = xd + att.permute(0, 2, 1, 3).reshape(B, Q, D) @ out_w.double().t() + out_b.double()
    assert (y.cpu().double() - ref).abs().max() < 2e-5


def _bf16x3(x, w, b, res=None, ln=None, splitk=0):
    import ctypes
    from sparsebev_amd import _lib
    lib = _lib.load()
    M, K = x.shape
    N = w.shape[0]
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    w2 = torch.empty(N, 2 * K, device=DEV, dtype=torch.int16)
    assert lib.sbev_split_bf16x3_weights(p(w), p(w2), N, K, st) == 0
    y = torch.empty(M, N, device=DEV)
    if splitk:
        ws = torch.empty(splitk, M, N, device=DEV)
        rc = lib.sbev_linear_splitk_bf16x3(p(x), p(w2), p(b), p(res), p(ln[0] if ln else None), p(ln[1] if ln else None), 1e-5,
                                           p(y), M, N, K, K, 0, splitk, p(ws), st)
    else:
        rc = lib.sbev_linear_bf16x3(p(x), p(w2), p(b), p(res), p(y), M, N, K, K, N, 0, st)
    assert rc == 0, lib.sbev_last_error()
    return y


@pytest.mark.parametrize('M,N,K,splitk', [(900, 32768, 256, 0), (900, 256, 32768, 32), (130, 200, 64, 0), (37, 256, 1024, 3)])
def test_bf16x3_linear_fp32_class_accuracy(M, N, K, splitk):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    y = _bf16x3(x, w, b, splitk=splitk)
    ref = x.double() @ w.double().t() + b.double()
    d = (y.double() - ref).abs()
    err, rms = d.max().item(), d.pow(2).mean().sqrt().item()
    plain = (x.to(torch.bfloat16).float() @ w.to(torch.bfloat16).float().t() + b - ref.float()).abs().max().item()
    # outputs are O(1) sums of K products, each good to ~3 * 2^-18 relative: rms error ~3e-6, extreme tail of up to 3e7
    # outputs a few 1e-5 -- fp32-class, two orders of magnitude better than a plain bf16 GEMM
    assert rms < 6e-6 and err < 6e-5, (rms, err)
    assert plain > 30 * err


@pytest.mark.parametrize('M', [1, 15, 16, 17, 33, 900, 1800, 3200, 3600])
@pytest.mark.parametrize('N,relu,use_bias', [(32768, False, True), (24576, True, True), (24576, False, False)])
def test_generator_strip_kernel_ragged_rows(M, N, relu, use_bias):
    """[M,256] x [N,256]^T with N/128 >= 192 goes to the W-stationary strip kernel (gemm_nt_f32_strip_kernel): every
    row-fragment remainder (M % 16, odd / even fragment counts for the two row-halves) against fp64."""
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, 256, generator=g)
    w = torch.randn(N, 256, generator=g) / 16
    b = torch.randn(N, generator=g) if use_bias else None
    y = dense.linear(x.to(DEV), w.to(DEV), b.to(DEV) if use_bias else None, relu=relu)
    ref = ref_linear(x, w, b, relu)
    assert y.shape == (M, N)
    assert (y.cpu().double() - ref).abs().max() < 2e-5


def test_linear_group_equals_single_launches():
    """sbev_linear_group_f32: the cls / reg branch levels launched side by side must give bit-identical results to
    one sbev_linear_f32 per problem (same tile arithmetic), for ragged M / N too."""
    g = torch.Generator().manual_seed(5)
    dev = 'cuda:0'
    for M, Ns, K in [(900, (256, 256), 256), (900, (10, 10), 256), (70, (256, 10, 33), 512), (1, (5,), 256)]:
        probs = []
        for i, N in enumerate(Ns):
            x = torch.randn(M, K, generator=g).to(dev)
            w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
            b = torch.randn(N, generator=g).to(dev) if i != 1 else None
            probs.append((x, w, b, i % 2 == 1))
        outs = dense.linear_group(probs)
        for (x, w, b, relu), y in zip(probs, outs):
            ref = dense.linear(x, w, b, relu=relu)
            assert torch.equal(y, ref), (M, Ns, K)
            t = x.double() @ w.double().T + (b.double() if b is not None else 0)
            t = t.clamp_min(0) if relu else t
            assert (y.double() - t).abs().max() < 1e-4


def test_linear_group_rejects_bad_groups():
    dev = 'cuda:0'
    x = torch.zeros(8, 256, device=dev); w = torch.zeros(4, 256, device=dev)
    x2 = torch.zeros(8, 512, device=dev); w2 = torch.zeros(4, 512, device=dev)
    with pytest.raises(RuntimeError):
        dense.linear_group([(x, w, None, False), (x2, w2, None, False)])      # mixed K
    with pytest.raises(RuntimeError):
        dense.linear_group([(x, w, None, False)] * 4)                           # more than 3
    x3 = torch.zeros(8, 128, device=dev); w3 = torch.zeros(4, 128, device=dev)
    with pytest.raises(RuntimeError):
        dense.linear_group([(x3, w3, None, False)])                             # K not 256 / 512


@pytest.mark.parametrize('M,N,ln_relu,with_add,relu', [(900, 776, True, True, False), (900, 112, False, False, False),
                                                        (900, 256, False, False, True), (37, 40, True, False, False), (3200, 128, False, True, False),
                                                        (3200, 776, True, True, False), (3600, 776, True, True, False), (3600, 112, False, False, False),
                                                        (2048, 256, False, False, True), (2049, 256, False, False, True)])
def test_layer_norm_as_linear_prologue(M, N, ln_relu, with_add, relu):
    # dense.ln_linear (one launch, row statistics exchanged through LDS inside the consumer's tiles) against the two
    # launches it replaces; the stored normalised rows against the stand-alone LayerNorm
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, 256, generator=g) * 3 + 0.7).to(DEV)
    lw, lb = (torch.rand(256, generator=g) + 0.5).to(DEV), (torch.randn(256, generator=g) * 0.1).to(DEV)
    w, b = (torch.randn(N, 256, generator=g) * 0.05).to(DEV), torch.randn(N, generator=g).to(DEV)
    add = torch.randn(M, 256, generator=g).to(DEV) if with_add else None
    xn, y = dense.ln_linear(x, lw, lb, w, b, ln_relu=ln_relu, add_after=add, relu=relu)
    xn_ref = dense.layer_norm(x, lw, lb, relu=ln_relu, add_after=add)
    y_ref = dense.linear(xn_ref, w, b, relu=relu)
    assert (xn - xn_ref).abs().max().item() < 5e-6
    assert (y - y_ref).abs().max().item() < 2e-5
    # and against plain torch on the host
    xc = torch.nn.functional.layer_norm(x.cpu().double(), (256,), lw.cpu().double(), lb.cpu().double(), 1e-5)
    if ln_relu:
        xc = xc.relu()
    if add is not None:
        xc = xc + add.cpu().double()
    yc = xc @ w.cpu().double().T + b.cpu().double()
    if relu:
        yc = yc.relu()
    assert (xn.cpu().double() - xc).abs().max().item() < 2e-5
    assert (y.cpu().double() - yc).abs().max().item() < 1e-4
    # the consumer of the stored rows sees exactly what the prologue fed the MFMAs
    assert torch.equal(dense.linear(xn, w, b, relu=relu), y)


def test_layer_norm_prologue_falls_back_outside_the_small_tile_shapes():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, 512, generator=g).to(DEV)                       # K = 512: LayerNorm + Linear as two launches
    lw, lb = torch.rand(512, generator=g).to(DEV), torch.randn(512, generator=g).to(DEV)
    w, b = (torch.randn(96, 512, generator=g) * 0.05).to(DEV), torch.randn(96, generator=g).to(DEV)
    xn, y = dense.ln_linear(x, lw, lb, w, b)
    xn_ref = dense.layer_norm(x, lw, lb)
    assert torch.equal(xn, xn_ref) and torch.equal(y, dense.linear(xn_ref, w, b))


@pytest.mark.parametrize('M,N,relu', [(900, 32768, 0), (3600, 20480, 0), (1, 1024, 1), (33, 2048, 1), (100, 18432, 0)])
def test_bf16x3_strip_generator_kernel(M, N, relu):
    """sbev_linear_bf16x3_strip (W-stationary strips, the activation split once): the same three bf16 products as the tile
    kernel in another summation order -- fp32-class accuracy against fp64 and agreement with the tile kernel to round-off."""
    import ctypes
    from sparsebev_amd import _lib
    lib = _lib.load()
    K = 256
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.sbev_linear_bf16x3_strip_ok(M, N, K) == 1 and lib.sbev_linear_bf16x3_strip_ok(M, N, 512) == 0
    w2 = torch.empty(N, 2 * K, device=DEV, dtype=torch.int16)
    x2 = torch.empty(M, 2 * K, device=DEV, dtype=torch.int16)
    assert lib.sbev_split_bf16x3_weights(p(w), p(w2), N, K, st) == 0 and lib.sbev_split_bf16x3_weights(p(x), p(x2), M, K, st) == 0
    y = torch.full((M, N), float('nan'), device=DEV)
    assert lib.sbev_linear_bf16x3_strip(p(x2), p(w2), p(b), p(y), M, N, K, N, relu, st) == 0, lib.sbev_last_error()
    ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    d = (y.double() - ref).abs()
    assert d.pow(2).mean().sqrt().item() < 6e-6 and d.max().item() < 6e-5
    tile = _bf16x3(x, w, b)
    if relu:
        tile = tile.clamp_min(0)
    assert (y - tile).abs().max().item() < 2e-5
    # refuses other shapes with the documented status
    assert lib.sbev_linear_bf16x3_strip(p(x2), p(w2), p(b), p(y), M, N, 512, N, relu, st) == -1
