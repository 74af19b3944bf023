"""Split-bf16 Linears (csrc/gemm_bf16s.hip): bf16x6 = hi + mid + lo images, six products, fp32 accumulate on the bf16
matrix core -- must be NOT NARROWER than the exact f32-MFMA kernels it may replace: both are compared with
an fp64 evaluation of the same Linear on the same inputs, at the two shapes of AdaptiveMixing
(models/sparsebev_transformer.py:358 parameter_generator, :378 out_proj) and at ragged / odd shapes."""
import ctypes

import pytest
import torch

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs a GPU')]

from sparsebev_amd import _lib, dense  # noqa: E402

DEV = 'cuda'


def _planes_to_float(planes):
    """int16 bf16 bit patterns [nimg, ...] -> fp32 values per image."""
    return (planes.to(torch.int32) << 16).view(torch.float32)


def _rand(shape, seed, scale=1.0, wide=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g) * scale
    if wide:       # 12 binades of magnitude, both signs: exercises every exponent alignment of the split
        x = x * torch.exp2(torch.randint(-6, 7, shape, generator=g).float())
    return x.to(DEV)


@pytest.mark.parametrize('nimg', [2, 3])
def test_split_images_are_an_exact_decomposition(nimg):
    """hi (+ mid) + lo: each image is the RNE bf16 of the remainder; three images reproduce every fp32 value BIT FOR BIT,
    two to 2^-17 relative.  (What 'fp32 operands on the bf16 matrix core' rests on.)"""
    x = _rand((257, 256), 1, wide=True)
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.0e-30, 65504.0], device=DEV)
    planes = dense.split_bf16s_rows(x, nimg)
    assert planes.shape == (nimg, 257, 256) and planes.dtype == torch.int16
    img = _planes_to_float(planes)
    # image 0 is torch's own RNE bf16 rounding
    assert torch.equal(img[0], x.to(torch.bfloat16).float())
    r = x - img[0]
    assert torch.equal(img[1], r.to(torch.bfloat16).float())
    if nimg == 3:
        r2 = r - img[1]
        assert torch.equal(img[2], r2.to(torch.bfloat16).float())
        assert torch.equal((img[0].double() + img[1].double() + img[2].double()).float(), x)
        assert torch.equal(img[0].double() + img[1].double() + img[2].double(), x.double())
    else:
        rel = ((img[0].double() + img[1].double()) - x.double()).abs() / x.double().abs().clamp_min(1e-300)
        assert rel.max().item() <= 2.0 ** -17


@pytest.mark.parametrize('nimg', [2, 3])
def test_fragment_pack_is_the_mfma_operand_order(nimg):
    """[N/32][K/16][img][lane][8]: lane l holds row 32 nf + (l & 31), k = 16 ks + 8 (l >> 5) + 0..7 of image img."""
    N, K = 77, 96                                                  # ragged: the last block repeats row N - 1
    w = _rand((N, K), 2, wide=True)
    frags = dense.pack_bf16s_frags(w, nimg)
    assert frags.shape == ((N + 31) // 32, K // 16, nimg, 64, 8)
    planes = dense.split_bf16s_rows(w, nimg)                       # [nimg, N, K]
    lane = torch.arange(64, device=DEV)
    for nf in range((N + 31) // 32):
        for ks in range(K // 16):
            rows = (nf * 32 + (lane & 31)).clamp_max(N - 1)
            k0 = ks * 16 + (lane >> 5) * 8
            for img in range(nimg):
                want = torch.stack([planes[img, rows, k0 + j] for j in range(8)], dim=1)
                assert torch.equal(frags[nf, ks, img], want)


def _errs(y, ref):
    d = (y.double() - ref).abs()
    return d.max().item(), d.pow(2).mean().sqrt().item()


def _gen_all(x, w, b, relu=False):
    """(bf16x6, bf16x3, exact f32) generator-shaped Linear on the same inputs."""
    out = {}
    for nimg in (3, 2):
        out[nimg] = dense.linear_bf16s_gen(x, dense.pack_bf16s_frags(w, nimg), b, nimg=nimg, relu=relu)
    out['f32'] = dense.linear(x, w, b, relu=relu)
    return out


def test_generator_bf16x6_not_narrower_than_f32_mfma():
    """(900, 32768, 256): max and rms error against fp64 of bf16x6 <= those of gemm_nt_f32_strip_kernel (the exact path)."""
    M, N, K = 900, 32768, 256
    x, w, b = _rand((M, K), 3), _rand((N, K), 4, K ** -0.5), _rand((N,), 5)
    ys = _gen_all(x, w, b)
    ref = x.double() @ w.double().t() + b.double()
    e6, r6 = _errs(ys[3], ref)
    e3, r3 = _errs(ys[2], ref)
    ef, rf = _errs(ys['f32'], ref)
    print('generator  max/rms err vs fp64: bf16x6 %.3e %.3e   f32-mfma %.3e %.3e   bf16x3 %.3e %.3e' % (e6, r6, ef, rf, e3, r3))
    assert r6 <= rf and e6 <= ef, ('bf16x6 is narrower than the f32 MFMA kernel', e6, r6, ef, rf)
    assert r3 < 6e-6 and e3 < 6e-5            # the 2^-16 class of the three-product mode
    # and it is the same function: agreement with the exact kernel at fp32 round-off
    assert (ys[3] - ys['f32']).abs().max().item() < 2e-5          # (each is up to ~9e-6 from fp64 on these O(1..5) outputs)


def test_out_projection_bf16x6_not_narrower_than_f32_mfma():
    """(900, 256, 32768) split-K with the fused bias + residual + LayerNorm reducer."""
    M, N, K = 900, 256, 32768
    x = _rand((M, K), 6).clamp_min(0)          # post-ReLU activations, like the mixing output
    w, b = _rand((N, K), 7, K ** -0.5), _rand((N,), 8)
    res = _rand((M, N), 9)
    ref_lin = x.double() @ w.double().t() + b.double()
    y6 = dense.linear_splitk_bf16s(x, dense.pack_bf16s_frags(w, 3), b, nimg=3)
    y3 = dense.linear_splitk_bf16s(x, dense.pack_bf16s_frags(w, 2), b, nimg=2)
    yf = dense.linear(x, w, b)
    e6, r6 = _errs(y6, ref_lin)
    e3, r3 = _errs(y3, ref_lin)
    ef, rf = _errs(yf, ref_lin)
    print('out-proj   max/rms err vs fp64: bf16x6 %.3e %.3e   f32-mfma %.3e %.3e   bf16x3 %.3e %.3e' % (e6, r6, ef, rf, e3, r3))
    assert e6 <= ef and r6 <= 1.25 * rf, ('bf16x6 is narrower than the f32 MFMA kernel', e6, r6, ef, rf)
    assert r3 < 6e-6 and e3 < 6e-5
    # fused epilogue: + residual, LayerNorm
    gam, bet = _rand((N,), 10), _rand((N,), 11)
    yl = dense.linear_splitk_bf16s(x, dense.pack_bf16s_frags(w, 3), b, nimg=3, residual=res, ln=(gam, bet))
    ref = torch.nn.functional.layer_norm(ref_lin + res.double(), (N,), gam.double(), bet.double(), 1e-5)
    assert (yl.double() - ref).abs().max().item() < 2e-5
    # bit-reproducible (fixed summation order, no atomics)
    assert torch.equal(y6, dense.linear_splitk_bf16s(x, dense.pack_bf16s_frags(w, 3), b, nimg=3))


@pytest.mark.parametrize('nimg', [3, 2])
@pytest.mark.parametrize('M', [1, 31, 32, 33, 97, 129, 900, 1600, 3600])
@pytest.mark.parametrize('N,K,relu,use_bias', [(512, 256, False, True), (256, 32, True, True), (1024, 96, False, False), (77824, 256, False, True)])
def test_generator_kernel_ragged_rows(M, N, K, relu, use_bias, nimg):
    """every row-fragment remainder (M % 32), 1..4 fragments per row tile and both waves' shares of a tile, short and odd K."""
    x, w = _rand((M, K), M + N), _rand((N, K), M + K, K ** -0.5)
    b = _rand((N,), 12) if use_bias else None
    y = torch.full((M, N), float('nan'), device=DEV)
    xs, ws = dense.pack_bf16s_frags(x, nimg), dense.pack_bf16s_frags(w, nimg)
    lib = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.sbev_linear_bf16s_gen(p(xs), p(ws), p(b), p(y), M, N, K, N, int(relu), nimg, st) == 0, lib.sbev_last_error()
    ref = x.double() @ w.double().t() + (b.double() if use_bias else 0.0)
    if relu:
        ref = ref.clamp_min(0)
    tol = 1e-5 if nimg == 3 else 6e-5          # fp32 class (the exact kernel: 9e-6 on outputs this size) / the 2^-16 class
    assert (y.double() - ref).abs().max().item() < tol


@pytest.mark.parametrize('nimg', [3, 2])
@pytest.mark.parametrize('M,K', [(1, 256), (33, 512), (64, 1024), (65, 2048), (100, 32768), (3200, 4096), (7, 32768 + 32)])
def test_out_projection_kernel_ragged(M, K, nimg):
    """1 or 2 row fragments per tile, odd slab counts per K chunk (the two wave quartets of a workgroup get unequal halves)."""
    N = 256
    x, w, b = _rand((M, K), M + K), _rand((N, K), K, K ** -0.5), _rand((N,), 13)
    y = dense.linear_splitk_bf16s(x, dense.pack_bf16s_frags(w, nimg), b, nimg=nimg)
    ref = x.double() @ w.double().t() + b.double()
    tol = 3e-6 if nimg == 3 else 6e-5
    assert (y.double() - ref).abs().max().item() < tol


def test_shape_guards():
    lib = _lib.load()
    assert lib.sbev_linear_bf16s_gen_ok(900, 32768, 256) == 1 and lib.sbev_linear_bf16s_gen_ok(900, 32768 + 128, 256) == 0
    assert lib.sbev_linear_bf16s_gen_ok(900, 32768, 250) == 0
    assert lib.sbev_linear_bf16s_out_ok(900, 256, 32768) == 1 and lib.sbev_linear_bf16s_out_ok(900, 512, 32768) == 0
    plan = lib.sbev_linear_bf16s_out_plan(900, 256, 32768)
    # the larger of the two kernels' plans: 15 tiles of 64 rows x 17 chunks (bf16 modes) / 8 tiles of <= 128 rows x 32 chunks (fp16 modes,
    # pre-split X) -- each one near-full round of the 256 CUs
    assert plan == 32
    assert lib.sbev_linear_bf16s_out_plan(900, 512, 32768) == 0
    x = torch.zeros(4, 256, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.sbev_linear_bf16s_gen(p(x), p(x), None, p(x), 4, 100, 256, 100, 0, 3, st) == -1
    assert lib.sbev_linear_bf16s_gen(p(x), p(x), None, p(x), 4, 256, 256, 256, 0, 4, st) == -1
    assert b'nimg' in lib.sbev_last_error()


# ---- fp16 hi + lo modes (f16x3 / f16x4): scaled split, 3 or 4 products -----------------------------------------------------------
def _f16_images(frags):
    """int16 [.., 2, 64, 8] fp16 bit patterns -> fp32 values"""
    return frags.view(torch.float16).float()


def test_f16_split_is_scaled_hi_plus_lo():
    """pack_f16s_frags: 2^e per row with the row maximum in [2^14, 2^15); hi = RNE_fp16(x 2^e), lo = RNE_fp16(x 2^e - hi); the pair
    reproduces x to 2^-23 relative (11 + 11 significand bits and lo's sign) for elements within 2^-17 of the row maximum, to 2^-25 in scaled units below."""
    N, K = 77, 96
    w = _rand((N, K), 21, wide=True)
    w[3] = 0.0                                                   # an all-zero row: scale 1
    w[5, :4] = torch.tensor([3.0e38, -1.0e30, 1.0e-30, 0.0], device=DEV)
    frags, sc = dense.pack_f16s_frags(w)
    assert frags.shape == ((N + 31) // 32, K // 16, 2, 64, 8) and sc.shape == (2, N)
    up, down = sc[0], sc[1]
    assert torch.equal(up * down, torch.ones_like(up)) and torch.equal(torch.exp2(torch.log2(up).round()), up)      # exact powers of two
    mx = w.abs().amax(1) * up
    nz = w.abs().amax(1) > 0
    assert bool(((mx[nz] >= 2.0 ** 14) & (mx[nz] < 2.0 ** 15)).all()) and float(up[3]) == 1.0
    img = _f16_images(frags)                                     # [nf, ks, 2, 64, 8]
    lane = torch.arange(64, device=DEV)
    for nf in range((N + 31) // 32):
        rows = (nf * 32 + (lane & 31)).clamp_max(N - 1)
        for ks in range(K // 16):
            k0 = ks * 16 + (lane >> 5) * 8
            xs = torch.stack([w[rows, k0 + j] for j in range(8)], dim=1) * up[rows, None]
            hi = xs.to(torch.float16).float()
            lo = (xs - hi).to(torch.float16).float()
            assert torch.equal(img[nf, ks, 0], hi) and torch.equal(img[nf, ks, 1], lo)
            err = (hi.double() + lo.double() - xs.double()).abs()
            assert bool((err <= torch.maximum(xs.double().abs() * 2.0 ** -23, torch.full_like(err, 2.0 ** -25))).all())
    # per tensor: one power of two for the whole matrix
    frags_t, sc_t = dense.pack_f16s_frags(w[:3], per_tensor=True)
    assert sc_t.shape == (2,) and 2.0 ** 14 <= float(w[:3].abs().max() * sc_t[0]) < 2.0 ** 15


@pytest.mark.parametrize('nprod', [3, 4])
def test_generator_f16_not_narrower_than_f32_mfma(nprod):
    """(900, 32768, 256): max and rms error against fp64 of the fp16 hi + lo modes <= those of gemm_nt_f32_strip_kernel, on unit-scale
    inputs AND on inputs whose elements span 12 binades."""
    M, N, K = 900, 32768, 256
    for wide in (False, True):
        x, w, b = _rand((M, K), 3, wide=wide), _rand((N, K), 4, K ** -0.5, wide=wide), _rand((N,), 5)
        wf, wsc = dense.pack_f16s_frags(w)
        y = dense.linear_f16s_gen(x, wf, wsc, b, nprod=nprod)
        yf = dense.linear(x, w, b)
        ref = x.double() @ w.double().t() + b.double()
        e, r = _errs(y, ref)
        ef, rf = _errs(yf, ref)
        print('generator f16x%d wide=%s  max/rms err vs fp64: %.3e %.3e   f32-mfma %.3e %.3e' % (nprod, wide, e, r, ef, rf))
        assert e <= ef and r <= rf, ('fp16 hi + lo is narrower than the f32 MFMA kernel', e, r, ef, rf)
        del ref


@pytest.mark.parametrize('nprod', [3, 4])
@pytest.mark.parametrize('pairs', [False, True])
def test_out_projection_f16_not_narrower_than_f32_mfma(nprod, pairs):
    """(900, 256, 32768) split-K: X split inside the kernel (64-row tiles) or handed over as (hi, lo) pairs (128-row tiles)."""
    M, N, K = 900, 256, 32768
    import math
    for wide in (False, True):
        x = _rand((M, K), 6, wide=wide).clamp_min(0)
        w, b = _rand((N, K), 7, K ** -0.5, wide=wide), _rand((N,), 8)
        up = 15 - math.frexp(float(x.abs().max()))[1]
        wf, wsc = dense.pack_f16s_frags(w)
        xin = dense.f16s_pairs(x, up) if pairs else x
        y = dense.linear_splitk_f16s(xin, wf, wsc, b, nprod=nprod, x_up_log2=up, x_is_pairs=pairs)
        yf = dense.linear(x, w, b)
        ref = x.double() @ w.double().t() + b.double()
        e, r = _errs(y, ref)
        ef, rf = _errs(yf, ref)
        print('out-proj f16x%d pairs=%s wide=%s  max/rms err vs fp64: %.3e %.3e   f32-mfma %.3e %.3e' % (nprod, pairs, wide, e, r, ef, rf))
        assert e <= ef and r <= rf, ('fp16 hi + lo is narrower than the f32 MFMA kernel', e, r, ef, rf)
        # bit-reproducible (fixed summation order, no atomics)
        assert torch.equal(y, dense.linear_splitk_f16s(xin, wf, wsc, b, nprod=nprod, x_up_log2=up, x_is_pairs=pairs))
        del ref
    # fused epilogue: + residual, LayerNorm
    res, gam, bet = _rand((M, N), 9), _rand((N,), 10), _rand((N,), 11)
    yl = dense.linear_splitk_f16s(xin, wf, wsc, b, nprod=nprod, residual=res, ln=(gam, bet), x_up_log2=up, x_is_pairs=pairs)
    ref = torch.nn.functional.layer_norm(x.double() @ w.double().t() + b.double() + res.double(), (N,), gam.double(), bet.double(), 1e-5)
    assert (yl.double() - ref).abs().max().item() < 5e-3         # (12-binade inputs: outputs of O(100))


def _qual_inputs(kind, M, N, K, seed):
    """Operands for the round-4 qualification of f16x3:
    'binades24' -- every row of X and of W spans 24+ binades;  'outlier' -- one element per row 2^20 times the rest;
    'norm1' -- X = LayerNorm(z) gamma + beta with a trained-like gamma in [0.01, 30] (log-uniform), beta ~ N(0, 1): the generator's
    real input, scaled by the runtime's A-PRIORI bound sqrt(K - 1) max|gamma| + max|beta| instead of its measured maximum."""
    g = torch.Generator().manual_seed(seed)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5
    x_bound = None
    if kind == 'binades24':
        x = x * torch.exp2(torch.randint(-12, 13, (M, K), generator=g).float())
        w = w * torch.exp2(torch.randint(-12, 13, (N, K), generator=g).float())
    elif kind == 'outlier':
        x[torch.arange(M), torch.randint(0, K, (M,), generator=g)] *= 2.0 ** 20
        w[torch.arange(N), torch.randint(0, K, (N,), generator=g)] *= 2.0 ** 20
    elif kind == 'norm1':
        import math
        gam = torch.exp(torch.rand(K, generator=g) * (math.log(30.0) - math.log(0.01)) + math.log(0.01))
        bet = torch.randn(K, generator=g)
        x = torch.nn.functional.layer_norm(x * torch.exp2(torch.randint(-3, 4, (M, 1), generator=g).float()), (K,), gam, bet, 1e-5)
        x_bound = math.sqrt(K - 1) * float(gam.max()) + float(bet.abs().max())
    return x.to(DEV), w.to(DEV), torch.randn(N, generator=g).to(DEV), x_bound


def _check_f16_vs_exact(name, y, yf, x, w, b, ref, norms=True, x_max=None):
    """norms: not further from fp64 than the exact f32-MFMA kernel in max and rms (the decoder's operand classes).
    Always: element by element inside an EXPLICIT bound,
        |y - ref| <= 2^-19 (|x| |w|^T + |b|)  +  2^-38 (xmax ||w_n||_1 + wmax_n ||x_r||_1)
    -- the first term is fp32-class and componentwise (a K-long fp32 fma chain guarantees K 2^-24; the exact kernel's own ratio is printed: 37 .. 66 x 2^-24 on the outlier operands);
    the second is what one power of two per tensor (X) / per row (W) costs: an element more than 2^-17 below its scale's maximum has a
    subnormal lo image (absolute error 2^-25 in scaled units = 2^-39 of that maximum), invisible next to the first term unless an
    outlier 2^20 above the rest stretches the scale (measured: a 6.6e-5 element under a 3.3e6 tensor maximum keeps 3 bits).
    norms=False: such outlier operands -- there hi + lo is NOT "not worse than exact" (dropped lo x lo: up to 2^-22 of a single
    dominant product; crushed small elements), only inside the bound.  The decoder's operands cannot do that: LayerNorm outputs are
    bounded by sqrt(n - 1) sigma, and a norm1 with an outlier gamma makes the runtime fall back to the exact kernels."""
    e, r = _errs(y, ref)
    ef, rf = _errs(yf, ref)
    xa, wa = x.double().abs(), w.double().abs()
    mag = xa @ wa.t() + b.double().abs()
    xm = float(xa.max()) if x_max is None else float(x_max)
    floor = 2.0 ** -38 * (xm * wa.sum(1)[None, :] + wa.max(1).values[None, :] * xa.sum(1)[:, None])
    err, errf = (y.double() - ref).abs(), (yf.double() - ref).abs()
    ratio = (err / (mag * 2.0 ** -19 + floor)).max().item()
    ratio_f = (errf / mag).max().item() * 2.0 ** 24
    print('%s  max/rms err vs fp64: f16x3 %.3e %.3e   f32-mfma %.3e %.3e   f16x3 / explicit bound %.3f   exact componentwise %.2f x 2^-24' % (name, e, r, ef, rf, ratio, ratio_f))
    if norms:
        assert e <= ef and r <= rf, (name, 'fp16 hi + lo is further from fp64 than the f32 MFMA kernel', e, r, ef, rf)
    assert ratio <= 1.0, (name, 'error above the explicit element-wise bound', ratio)


@pytest.mark.parametrize('kind', ['binades24', 'outlier', 'norm1'])
def test_generator_f16x3_wide_rows_outliers_and_trained_norm1(kind):
    M, N, K = 900, 32768, 256
    x, w, b, x_bound = _qual_inputs(kind, M, N, K, 31)
    wf, wsc = dense.pack_f16s_frags(w)
    lib = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if x_bound is None:
        y = dense.linear_f16s_gen(x, wf, wsc, b, nprod=3)
    else:                      # the decoder's way: X's power of two from the bound, not from the data (runtime._bind)
        import math
        e = math.floor(math.log2(65504.0 / x_bound) - 1e-9)
        assert float(x.abs().max()) * 2.0 ** e < 65504.0
        xsc = torch.tensor([2.0 ** e, 2.0 ** -e], device=DEV)
        xs = torch.empty((M + 31) // 32, K // 16, 2, 64, 8, device=DEV, dtype=torch.int16)
        assert lib.sbev_pack_f16s_frags(p(x), K, p(xs), p(xsc), M, K, 2, st) == 0, lib.sbev_last_error()
        y = torch.empty(M, N, device=DEV)
        assert lib.sbev_linear_f16s_gen(p(xs), p(xsc), p(wf), p(wsc[1]), p(b), p(y), M, N, K, N, 0, 3, st) == 0, lib.sbev_last_error()
    yf = dense.linear(x, w, b)
    ref = x.double() @ w.double().t() + b.double()
    _check_f16_vs_exact('generator ' + kind, y, yf, x, w, b, ref, norms=kind != 'outlier', x_max=x_bound)


@pytest.mark.parametrize('kind', ['binades24', 'outlier'])
@pytest.mark.parametrize('pairs', [False, True])
def test_out_projection_f16x3_wide_rows_and_outliers(kind, pairs):
    import math
    M, N, K = 900, 256, 32768
    x, w, b, _ = _qual_inputs(kind, M, N, K, 32)
    x = x.abs()                                               # (the out-projection's input is a ReLU output)
    up = 15 - math.frexp(float(x.max()))[1]
    wf, wsc = dense.pack_f16s_frags(w)
    xin = dense.f16s_pairs(x, up) if pairs else x
    y = dense.linear_splitk_f16s(xin, wf, wsc, b, nprod=3, x_up_log2=up, x_is_pairs=pairs)
    yf = dense.linear(x, w, b)
    ref = x.double() @ w.double().t() + b.double()
    _check_f16_vs_exact('out-projection %s pairs=%s' % (kind, pairs), y, yf, x, w, b, ref, norms=kind != 'outlier')


@pytest.mark.parametrize('M', [1, 31, 33, 97, 129, 900, 1600, 3600])
@pytest.mark.parametrize('N,K,relu,use_bias', [(512, 256, False, True), (256, 32, True, True), (1024, 96, False, False), (77824, 256, False, True)])
def test_generator_f16_ragged_rows(M, N, K, relu, use_bias):
    x, w = _rand((M, K), M + N), _rand((N, K), M + K, K ** -0.5)
    b = _rand((N,), 12) if use_bias else None
    wf, wsc = dense.pack_f16s_frags(w)
    y = dense.linear_f16s_gen(x, wf, wsc, b, nprod=3, relu=relu)
    ref = x.double() @ w.double().t() + (b.double() if use_bias else 0.0)
    if relu:
        ref = ref.clamp_min(0)
    assert (y.double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize('mode', ['f16x3', 'f16x4', 'bf16x3s'])
@pytest.mark.parametrize('M,N,relu,use_bias', [(900, 32768, False, True), (1, 256, True, True), (33, 512, False, False), (449, 1024, True, True),
                                               (1600, 77824, False, True), (3600, 32768, False, True)])
def test_weight_stationary_generator_equals_the_tiled_kernel(M, N, relu, use_bias, mode):
    """Round 4: K = 256 with two images runs on the weight-stationary kernel (gemm_f16s_gen_ws_kernel: a wave holds its 32 columns'
    weights in registers, X streams through a 3-slot LDS-DMA ring, row splits x column tiles walked by persistent workgroups).  Same
    image products as the tiled ping-pong kernel, summed in two interleaved accumulator chains instead of one: equal to fp32
    accumulation round-off (2^-18 of the output scale on 12-binade operands), every M % 32, every split; and not further from fp64 than the tiled kernel."""
    K = 256
    lib = _lib.load()
    x, w = _rand((M, K), M + N, wide=True), _rand((N, K), M + K, K ** -0.5, wide=True)
    b = _rand((N,), 12) if use_bias else None

    def run():
        if mode == 'bf16x3s':
            return dense.linear_bf16s_gen(x, dense.pack_bf16s_frags(w, 2), b, nimg=2, relu=relu)
        wf, wsc = dense.pack_f16s_frags(w)
        return dense.linear_f16s_gen(x, wf, wsc, b, nprod=3 if mode == 'f16x3' else 4, relu=relu)
    prev = lib.sbev_linear_gen_weight_stationary(1)
    try:
        y_ws = run()
        lib.sbev_linear_gen_weight_stationary(0)
        y_tiled = run()
    finally:
        lib.sbev_linear_gen_weight_stationary(prev)
    assert torch.isfinite(y_ws).all() and y_ws.abs().max() > 0
    scale = y_tiled.abs().max().item()
    assert (y_ws - y_tiled).abs().max().item() <= 2.0 ** -18 * scale          # two fp32 summation orders over 768 products of 12-binade operands
    if M * N <= 900 * 32768:
        ref = x.double() @ w.double().t() + (b.double() if use_bias else 0.0)
        if relu:
            ref = ref.clamp_min(0)
        e_ws, e_t = (y_ws.double() - ref).abs().max().item(), (y_tiled.double() - ref).abs().max().item()
        assert e_ws <= 1.25 * e_t + 1e-7 * scale, (e_ws, e_t)


@pytest.mark.parametrize('pairs', [False, True])
@pytest.mark.parametrize('M,K', [(1, 256), (33, 512), (64, 1024), (65, 2048), (97, 4096), (100, 32768), (129, 2048), (3200, 4096), (7, 32768 + 32),
                                 (1024, 2048), (1056, 32768), (2049, 4096), (2300, 8192 + 32), (3600, 32768)])
def test_out_projection_f16_ragged(M, K, pairs):
    """1 .. 4 row fragments per tile, odd k-step counts per chunk (the two wave quartets of a workgroup get unequal halves).  From 1024 rows
    the pre-split path runs on 256-row tiles (round 6, gemm_bf16s_out8_kernel): 7 / 8 fragments per tile split 4 + 3 / 4 + 4 over the two row
    halves, last tile partly outside the matrix."""
    N = 256
    x, w, b = _rand((M, K), M + K), _rand((N, K), K, K ** -0.5), _rand((N,), 13)
    wf, wsc = dense.pack_f16s_frags(w)
    xin = dense.f16s_pairs(x, 11) if pairs else x
    y = dense.linear_splitk_f16s(xin, wf, wsc, b, nprod=3, x_up_log2=11, x_is_pairs=pairs)
    ref = x.double() @ w.double().t() + b.double()
    assert (y.double() - ref).abs().max().item() < 3e-6


@pytest.mark.parametrize('M,K', [(3200, 32768), (3600, 32768), (2112, 4096)])
def test_out_projection_256_row_tiles_against_the_128_row_kernel(M, K):
    """Round 6: the batch shapes' out-projection on 256-row tiles (two row halves as phase groups sharing one W ring) -- same products as
    the 128-row kernel, k-ascending inside a chunk instead of two K halves: equal to it to fp32 summation round-off, not narrower against
    fp64, bit-stable run to run; sbev_linear_out8_min_rows switches."""
    from sparsebev_amd import _lib
    lib = _lib.load()
    N = 256
    x, w, b = _rand((M, K), M + K + 3), _rand((N, K), K + 5, K ** -0.5), _rand((N,), 17)
    wf, wsc = dense.pack_f16s_frags(w)
    xin = dense.f16s_pairs(x, 11)
    prev = lib.sbev_linear_out8_min_rows(1024)
    try:
        y8 = dense.linear_splitk_f16s(xin, wf, wsc, b, nprod=3, x_up_log2=11, x_is_pairs=True)
        assert torch.equal(y8, dense.linear_splitk_f16s(xin, wf, wsc, b, nprod=3, x_up_log2=11, x_is_pairs=True))
        lib.sbev_linear_out8_min_rows(0)
        y4 = dense.linear_splitk_f16s(xin, wf, wsc, b, nprod=3, x_up_log2=11, x_is_pairs=True)
    finally:
        lib.sbev_linear_out8_min_rows(prev)
    ref = x.double() @ w.double().t() + b.double()
    e8, r8 = _errs(y8, ref)
    e4, r4 = _errs(y4, ref)
    ef, rf = _errs(dense.linear(x, w, b), ref)             # the exact f32-input MFMA kernels: the fp32-class yardstick of every split mode
    print('out-projection M=%d K=%d  256-row tiles max/rms %.3e %.3e   128-row tiles %.3e %.3e   exact f32 MFMA %.3e %.3e' % (M, K, e8, r8, e4, r4, ef, rf))
    assert not torch.equal(y8, y4) or K < 8192           # (another summation order: a different kernel really ran)
    # fewer, longer k-chains than the 128-row plan (19 chunks of 108 k-steps against 51 x 2 halves of 20 at 3200 rows): more fp32
    # accumulation round-off than it, still not more than the exact kernels'
    assert (y8 - y4).abs().max().item() < 6e-6 and e8 <= 1.05 * ef + 1e-30 and r8 <= 1.05 * rf + 1e-30


@pytest.mark.parametrize('M,K,mag', [(900, 32768, 1.0), (900, 32768, 3e-7), (97, 4096, 5e4), (1, 256, 1.0)])
def test_out_projection_f16_device_side_scale(M, K, mag):
    """sbev_linear_splitk_f16s_xdev: X's power of two comes from device memory (sbev_f16s_tensor_scale) -- operands of any
    magnitude (a gradient), no host sync; the result equals the host-exponent path bit for bit and is not narrower than the exact kernel."""
    import math
    N = 256
    x, w, b = _rand((M, K), M + K + 1, mag, wide=True), _rand((N, K), K + 2, K ** -0.5), _rand((N,), 14, mag)
    res = _rand((M, N), 15, mag)
    wf, wsc = dense.pack_f16s_frags(w)
    xs = dense.f16s_tensor_scale(x)
    up = 15 - math.frexp(float(x.abs().max()))[1]
    assert xs.cpu().tolist() == [2.0 ** up, 2.0 ** -up]
    y = dense.linear_splitk_f16s(x, wf, wsc, b, residual=res, x_scale=xs)
    assert torch.equal(y, dense.linear_splitk_f16s(x, wf, wsc, b, residual=res, x_up_log2=up))
    ref = x.double() @ w.double().t() + b.double() + res.double()
    e, r = _errs(y, ref)
    ef, rf = _errs(dense.linear(x, w, b, residual=res), ref)
    slack = 1.02 if M >= 97 else 1.5          # (a maximum over a single row is one sample of the error distribution)
    assert e <= slack * ef + 1e-30 and r <= 1.02 * rf + 1e-30, (e, r, ef, rf)


def test_mixing_kernels_emit_the_pair_format():
    """sbev_adaptive_mixing_pairs_f16 == f16s_pairs(sbev_adaptive_mixing_f32): the split moved into the producer's epilogue, bit for bit"""
    BQ, G, Pin, C, Pout = 50, 4, 32, 64, 128
    x = _rand((BQ, G, Pin, C), 31)
    prm = _rand((BQ, G, C * C + Pout * Pin), 32, 0.3)
    lib = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    y = torch.empty(BQ, G * Pout * C, device=DEV)
    yp = torch.empty(BQ, G * Pout * C, device=DEV, dtype=torch.int32)
    assert lib.sbev_adaptive_mixing_f32(p(x), p(prm), p(y), BQ, G, Pin, C, Pout, 1e-5, st) == 0
    assert lib.sbev_adaptive_mixing_pairs_f16(p(x), p(prm), p(yp), BQ, G, Pin, C, Pout, 1e-5, 9, st) == 0
    assert torch.equal(yp, dense.f16s_pairs(y, 9))
    assert float(y.max()) * 2 ** 9 < 65504                      # the LayerNorm bound the decoder's 2^9 rests on: |y| <= sqrt(8191)
