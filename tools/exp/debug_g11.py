"""Dev tool: where do the decoder gradients of the HIP path and the fp32 / fp64 CPU oracle differ (per tensor, per layer count)?"""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sparsebev_amd import synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer
from oracle import sparsebev_oracle as O
DEV = 'cuda:0'
PREFIX = 'decoder.decoder_layer.'

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()

def run(n_layers, B=2, Q=36, T=2, L=4, pyr='tiny', dtype=torch.float64, feat_grad=True, qseed=111):
    ih, iw, sizes = S.PYRAMIDS[pyr]
    params = S.make_params(11, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=n_layers, num_levels=L, pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    bbox, feat = S.make_queries(B, Q, seed=qseed)
    feats = S.make_features(B, T, sizes, seed=112)
    metas = S.make_img_metas(B, T, ih, iw)
    g = torch.Generator().manual_seed(113)
    cc, cb = torch.randn(n_layers, B, Q, 10, generator=g), torch.randn(n_layers, B, Q, 10, generator=g)
    bd, fd = bbox.to(DEV).requires_grad_(True), feat.to(DEV).requires_grad_(True)
    ftd = [f.to(DEV).requires_grad_(feat_grad) for f in feats]
    cls, box = m(bd, fd, list(ftd), None, copy.deepcopy(metas))
    ((cls * cc.to(DEV)).sum() + (box * cb.to(DEV)).sum()).backward()
    po = {k: v.to(dtype).requires_grad_(True) for k, v in params.items()}
    bo, fo = bbox.to(dtype).requires_grad_(True), feat.to(dtype).requires_grad_(True)
    fto = [f.to(dtype).requires_grad_(True) for f in feats]
    metas_o = copy.deepcopy(metas)
    import numpy as np
    clo, boo, _ = O.decoder(po, bo, fo, fto, metas_o, S.PC_RANGE, num_layers=n_layers)
    ((clo * cc.to(dtype)).sum() + (boo * cb.to(dtype)).sum()).backward()
    errs = {'out_cls': rel(cls, clo), 'out_box': rel(box, boo), 'g.query_feat': rel(fd.grad, fo.grad), 'g.query_bbox': rel(bd.grad, bo.grad)}
    for (k, p) in m.named_parameters():
        errs[k[len(PREFIX):]] = rel(p.grad, po[k[len(PREFIX):]].grad)
    for i, (a, b) in enumerate(zip(ftd, fto)):
        if feat_grad:
            errs['feat%d' % i] = rel(a.grad, b.grad)
    print('layers', n_layers, 'oracle', dtype, 'B', B, 'feat_grad', feat_grad, 'qseed', qseed)
    for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:14]:
        print('   %-40s %.3e' % (k, v))

with torch.enable_grad():
    for kw in (dict(B=2), dict(B=1), dict(B=1, feat_grad=False), dict(B=2, feat_grad=False), dict(B=1, qseed=12), dict(B=2, qseed=12)):
        try:
            run(1, **kw)
        except Exception as e:
            print('fail', kw, repr(e)[:300])
