"""GPU tests of the row-chain kernel (csrc/rowchain.hip, sbev_row_chain) against fp64 torch math: every op kind,
ragged row counts, column counts that are not multiples of the 64-column pass, and the chains the decoder uses."""
import pytest
import torch
import torch.nn.functional as F

from sparsebev_amd import dense
from sparsebev_amd.dense import CHAIN_LINEAR, CHAIN_LINEAR3, CHAIN_LOAD, CHAIN_REFINE, chain_op, row_chain

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rnd(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def ln64(x, w, b):
    return F.layer_norm(x.double(), [x.shape[-1]], w.double(), b.double())


@pytest.mark.parametrize('M', [1, 16, 17, 900])
@pytest.mark.parametrize('K,N', [(256, 256), (256, 512), (512, 256), (256, 776), (256, 112), (256, 10), (128, 64)])
def test_single_linear_all_shapes(M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    x, w, b = rnd(g, M, K), rnd(g, N, K, scale=K ** -0.5), rnd(g, N)
    res = rnd(g, M, N)
    y = torch.empty(M, N, device=DEV)
    ops = [chain_op(CHAIN_LOAD, x, K, dst=0),
           chain_op(CHAIN_LINEAR, w, N, K=K, src=0, dst=1, bias=b, relu=True, res_g=res, out_g=y, to_lds=False)]
    row_chain(ops, M)
    ref = (x.double() @ w.double().t() + b.double()).clamp_min(0) + res.double()
    assert (y.double() - ref).abs().max() < 2e-5


def test_ffn_branches_refine_chain_vs_fp64():
    """The layer-tail chain: FFN (+ residual, LayerNorm) -> cls branch (Linear-LN-ReLU x2, Linear) and reg branch
    (Linear-ReLU x2, Linear) -> refine_bbox -> next layer's position encoder (Linear3-LN-ReLU, Linear-LN-ReLU, + x)
    -> attention in-projection; every intermediate that leaves the chain is checked."""
    g = torch.Generator().manual_seed(11)
    M, D, FF, NC, Q = 150, 256, 512, 10, 75
    x2 = rnd(g, M, D)
    bbox = torch.rand(M, 10, generator=g).to(DEV)
    vel_div = (torch.rand(M // Q, generator=g) + 0.5).to(DEV)
    W = lambda n, k: rnd(g, n, k, scale=k ** -0.5)
    V = lambda n: rnd(g, n, scale=0.3)
    G_ = lambda n: (1 + 0.1 * torch.randn(n, generator=g)).to(DEV)
    f0w, f0b, f1w, f1b, n3g, n3b = W(FF, D), V(FF), W(D, FF), V(D), G_(D), V(D)
    c0w, c0b, c1g, c1b, c3w, c3b, c4g, c4b, c6w, c6b = W(D, D), V(D), G_(D), V(D), W(D, D), V(D), G_(D), V(D), W(NC, D), V(NC)
    r0w, r0b, r2w, r2b, r4w, r4b = W(D, D), V(D), W(D, D), V(D), W(10, D), V(10)
    p0w, p0b, p1g, p1b, p3w, p3b, p4g, p4b = W(D, 3), V(D), G_(D), V(D), W(D, D), V(D), G_(D), V(D)
    iw, ib = W(776, D), V(776)
    x3, cls, box, xq, qkvt = [torch.empty(M, n, device=DEV) for n in (D, NC, 10, D, 776)]
    ops = [
        chain_op(CHAIN_LOAD, x2, D, dst=0),
        chain_op(CHAIN_LINEAR, f0w, FF, K=D, src=0, dst=1, bias=f0b, relu=True),
        chain_op(CHAIN_LINEAR, f1w, D, K=FF, src=1, dst=2, bias=f1b, res_buf=0, ln=(n3g, n3b), out_g=x3),
        chain_op(CHAIN_LINEAR, c0w, D, K=D, src=2, dst=0, bias=c0b, ln=(c1g, c1b), ln_relu=True),
        chain_op(CHAIN_LINEAR, c3w, D, K=D, src=0, dst=1, bias=c3b, ln=(c4g, c4b), ln_relu=True),
        chain_op(CHAIN_LINEAR, c6w, NC, K=D, src=1, dst=0, bias=c6b, out_g=cls, to_lds=False),
        chain_op(CHAIN_LINEAR, r0w, D, K=D, src=2, dst=0, bias=r0b, relu=True),
        chain_op(CHAIN_LINEAR, r2w, D, K=D, src=0, dst=1, bias=r2b, relu=True),
        chain_op(CHAIN_LINEAR, r4w, 12, K=D, src=1, dst=0, bias=None),          # placeholder replaced below
    ]
    ops[-1] = chain_op(CHAIN_LINEAR, torch.cat([r4w, torch.zeros(2, D, device=DEV)]), 12, K=D, src=1, dst=0,
                       bias=torch.cat([r4b, torch.zeros(2, device=DEV)]))       # N % 4 == 0 to stay in LDS
    ops += [
        chain_op(CHAIN_REFINE, bbox, 10, src=0, dst=1, out_g=box, aux=vel_div, aux_i=Q),
        chain_op(CHAIN_LINEAR3, p0w, D, src=1, dst=0, bias=p0b, ln=(p1g, p1b), ln_relu=True),
        chain_op(CHAIN_LINEAR, p3w, D, K=D, src=0, dst=1, bias=p3b, ln=(p4g, p4b), ln_relu=True, add_buf=2, out_g=xq),
        chain_op(CHAIN_LINEAR, iw, 776, K=D, src=1, dst=0, bias=ib, out_g=qkvt, to_lds=False),
    ]
    keep = [ops]                                                 # tensors referenced by raw pointers stay alive in this frame
    row_chain(ops, M)
    d = lambda t: t.double()
    h = (d(x2) @ d(f0w).t() + d(f0b)).clamp_min(0)
    x3r = ln64(d(x2) + h @ d(f1w).t() + d(f1b), n3g, n3b)
    c = ln64(x3r @ d(c0w).t() + d(c0b), c1g, c1b).clamp_min(0)
    c = ln64(c @ d(c3w).t() + d(c3b), c4g, c4b).clamp_min(0)
    clsr = c @ d(c6w).t() + d(c6b)
    r = (x3r @ d(r0w).t() + d(r0b)).clamp_min(0)
    r = (r @ d(r2w).t() + d(r2b)).clamp_min(0)
    reg = r @ d(r4w).t() + d(r4b)
    p = d(bbox)[:, :3].clamp(0, 1)
    xyz = torch.sigmoid(reg[:, :3] + torch.log(p.clamp_min(1e-5) / (1 - p).clamp_min(1e-5)))
    boxr = torch.cat([xyz, reg[:, 3:]], 1)
    boxr[:, 8:] = boxr[:, 8:] / d(vel_div).repeat_interleave(Q)[:, None]
    pos = ln64(boxr[:, :3] @ d(p0w).t() + d(p0b), p1g, p1b).clamp_min(0)
    pos = ln64(pos @ d(p3w).t() + d(p3b), p4g, p4b).clamp_min(0)
    xqr = x3r + pos
    qk = xqr @ d(iw).t() + d(ib)
    for name, got, ref in (('x3', x3, x3r), ('cls', cls, clsr), ('box', box, boxr), ('x', xq, xqr), ('qkvt', qkvt, qk)):
        assert (d(got) - ref).abs().max() < 1e-4, name
    assert keep


def test_row_chain_rejects_bad_programs():
    x = torch.zeros(4, 256, device=DEV)
    w = torch.zeros(256, 256, device=DEV)
    with pytest.raises(RuntimeError):
        row_chain([chain_op(CHAIN_LINEAR, w, 256, K=256, src=0, dst=0)], 4)              # in place
    with pytest.raises(RuntimeError):
        row_chain([chain_op(CHAIN_LINEAR, w, 256, K=200, src=0, dst=1)], 4)              # K not a multiple of 128
    with pytest.raises(RuntimeError):
        row_chain([chain_op(CHAIN_LOAD, x, 256, dst=3)], 4)                              # no such LDS buffer
    with pytest.raises(RuntimeError):
        row_chain([chain_op(CHAIN_LOAD, x, 256, dst=0)] * 15, 4)                         # too many ops
