"""Dev tool: is the 6-layer value-forced gradient of fixture G11-L6 well conditioned?  Compares (a) the fixture (fp32 reference,
CPU) and (b) the HIP path against the SAME value-forced chain evaluated by the oracle in fp64."""
import sys, os, copy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from conftest import load_golden
from sparsebev_amd import synthetic as S
from oracle import sparsebev_oracle as O
import test_gpu_backward as TB

g = load_golden('g11_train_L6')
B, Q, T, L, n_layers = [int(v) for v in g['cfg']]
seeds = [int(v) for v in g['seeds']]
ih, iw, sizes = S.PYRAMIDS[str(g['pyramid'])]
dt = torch.float64
params = {k: v.to(dt).requires_grad_(True) for k, v in S.make_params(seeds[0], embed_dims=256, num_frames=T, num_points=4, num_levels=L).items()}
feats = [f.to(dt).requires_grad_(True) for f in S.make_features(B, T, sizes, seed=seeds[2])]
metas = S.make_img_metas(B, T, ih, iw)
for b, m in enumerate(metas):
    m['img_timestamp'] = [float(v) for v in g['timestamps'][b]]
bbox, feat = g['query_bbox'].to(dt).requires_grad_(True), g['query_feat'].to(dt).requires_grad_(True)
with torch.enable_grad():
    fr, td, l2i, ih_, iw_ = O.decoder_prologue(feats, metas, O.msmv_sampling_gridsample)
    qb, qf = bbox, feat
    loss = 0
    for i in range(n_layers):
        qf, cls, box = O.decoder_layer(params, qb, qf, fr, td.to(dt), l2i.to(dt), ih_, iw_, S.PC_RANGE, T, 4, L, O.msmv_sampling_gridsample)
        loss = loss + (cls * g['cot_cls'][i].to(dt)).sum() + (box * g['cot_box'][i].to(dt)).sum()
        qb = g['out_bbox'][i].to(dt)
        qf = qf + (g['out_feat'][i].to(dt) - qf).detach()
    loss.backward()
def rel(a, b): return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()
errs_fix = {}
for name, gr in [(k, v.grad) for k, v in params.items()] + [('feat%d' % i, f.grad) for i, f in enumerate(feats)]:
    idx = S.grad_sample_indices(gr.numel())
    have = gr.reshape(g['g.' + name].shape) if idx is None else gr.reshape(-1)[idx]
    errs_fix[name] = rel(g['g.' + name], have)
print('fixture (fp32 reference) vs fp64 value-forced oracle: worst', sorted(errs_fix.items(), key=lambda kv: -kv[1])[:5], 'median %.2e' % sorted(errs_fix.values())[len(errs_fix) // 2])
if torch.cuda.is_available():
    with torch.enable_grad():
        errs, _ = TB._g11_run('L6', value_forced=True)      # HIP vs fixture
    print('HIP vs fixture: worst', sorted(errs.items(), key=lambda kv: -kv[1])[:5], 'median %.2e' % sorted(errs.values())[len(errs) // 2])
