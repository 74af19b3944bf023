"""Dev tool: SasaCore forward/backward vs an fp64 dense reference at the REAL operating points of the 6-layer G11 fixture."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch, torch.nn.functional as F
from conftest import load_golden
from sparsebev_amd import autograd as AG, synthetic as S
import test_gpu_backward as TB
DEV='cuda:0'
g = load_golden('g11_train_L6')
B, Q, T, L, n_layers = [int(v) for v in g['cfg']]
params = S.make_params(int(g['seeds'][0]), embed_dims=256, num_frames=T, num_points=4, num_levels=L)
ins = [(g['query_bbox'], g['query_feat'])] + [(g['out_bbox'][i], g['out_feat'][i]) for i in range(n_layers - 1)]
def ln(x, n): return F.layer_norm(x, [256], params[n + '.weight'].double(), params[n + '.bias'].double())
with torch.enable_grad():
  for i, (bbox, feat) in enumerate(ins):
    pos = bbox[..., :3].double()
    pos = ln(F.linear(pos, params['position_encoder.0.weight'].double(), params['position_encoder.0.bias'].double()), 'position_encoder.1').relu()
    pos = ln(F.linear(pos, params['position_encoder.3.weight'].double(), params['position_encoder.3.bias'].double()), 'position_encoder.4').relu()
    x = feat.double() + pos
    w = torch.cat([params['self_attn.attention.attn.in_proj_weight'], params['self_attn.gen_tau.weight']]).double()
    b = torch.cat([params['self_attn.attention.attn.in_proj_bias'], params['self_attn.gen_tau.bias']]).double()
    qkvt = F.linear(x, w, b).float()
    gy = torch.randn(B, Q, 256, generator=torch.Generator().manual_seed(i))
    qd = qkvt.to(DEV).requires_grad_(True)
    y = AG.SasaCore.apply(qd, bbox.to(DEV), None, tuple(S.PC_RANGE), 8, 0.0, 0)
    y.backward(gy.to(DEV))
    qc = qkvt.double().requires_grad_(True)
    yc = TB._sasa_ref(qc, bbox, None, 8)
    yc.backward(gy.double())
    D = 256
    tau = qkvt[..., 3 * D:]
    print('layer %d: fwd rel %.2e  grad rel %.2e  (dq %.2e dk %.2e dv %.2e dtau %.2e)  |tau| max %.2f  logits span %.1f' % (
        i, TB.rel(y, yc), TB.rel(qd.grad, qc.grad), TB.rel(qd.grad[..., :D], qc.grad[..., :D]), TB.rel(qd.grad[..., D:2*D], qc.grad[..., D:2*D]),
        TB.rel(qd.grad[..., 2*D:3*D], qc.grad[..., 2*D:3*D]), TB.rel(qd.grad[..., 3*D:], qc.grad[..., 3*D:]), tau.abs().max().item(), (qkvt[..., :D].abs().max() * qkvt[..., D:2*D].abs().max()).item()))
