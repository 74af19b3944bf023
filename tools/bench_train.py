"""Dev tool: time one TRAINING step of the decoder (forward + backward through all 6 layers, HIP kernels both ways) at a
bench config, features requiring grad or frozen.  Not the headline metric (that is inference samples/s, bench.py)."""
import argparse, copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsebev_amd import synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer

ap = argparse.ArgumentParser()
ap.add_argument('--q', type=int, default=900)
ap.add_argument('--t', type=int, default=8)
ap.add_argument('--b', type=int, default=1)
ap.add_argument('--pyr', default='r50_704x256')
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--feat-grad', action='store_true')
ap.add_argument('--dropout', action='store_true')
ap.add_argument('--graph', action='store_true', help='also time the step captured as ONE hipGraph (sparsebev_amd.train_graph.CapturedTrainStep; needs dropout off)')
ap.add_argument('--recompute', action='store_true', help='re-run generator GEMM + mixing in backward (the reference\'s checkpoint policy) instead of keeping 236 MB per layer')
a = ap.parse_args()
dev = 'cuda:0'
ih, iw, sizes = S.PYRAMIDS[a.pyr]
m = SparseBEVTransformer(256, num_frames=a.t, num_points=4, num_layers=6, num_levels=len(sizes), pc_range=S.PC_RANGE)
m.init_weights(); S.randomize_zero_init(m, std=0.02, seed=0)
m = m.to(dev).train()
m.decoder.decoder_layer.recompute_mixing = a.recompute
if not a.dropout:
    m.decoder.decoder_layer.self_attn.attn_drop = 0.0
    m.decoder.decoder_layer.ffn_drop = 0.0
feats = [f.requires_grad_(a.feat_grad) for f in S.make_features(a.b, a.t, sizes, seed=0, device=dev)]
bbox, feat = [t.to(dev) for t in S.make_queries(a.b, a.q, seed=0)]
feat.requires_grad_(True)
metas = S.make_img_metas(a.b, a.t, ih, iw)

def step():
    for p in m.parameters():
        p.grad = None
    cls, box = m(bbox, feat, list(feats), None, copy.deepcopy(metas))
    (cls.sum() + box.sum()).backward()

dg = None
if a.graph:
    # first thing in the process: the captured step wants the parameters' AccumulateGrad nodes on ITS stream (train_graph.py)
    from sparsebev_amd.train_graph import CapturedTrainStep
    cap = CapturedTrainStep(m, bbox, feat, feats, metas, lambda cls, box: cls.sum() + box.sum())
    for _ in range(3):
        cap.replay(metas)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cap.replay(metas)
    torch.cuda.synchronize()
    dg = (time.perf_counter() - t0) / a.steps
    del cap

for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
with torch.no_grad():
    m.eval()
    for _ in range(3):
        m(bbox, feat, list(feats), None, copy.deepcopy(metas))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m(bbox, feat, list(feats), None, copy.deepcopy(metas))
    torch.cuda.synchronize()
    di = (time.perf_counter() - t0) / a.steps
print('train step (fwd+bwd, 6 layers, Q=%d T=%d B=%d, feat_grad=%s, dropout=%s, recompute=%s): %.2f ms%s   inference step: %.2f ms   peak mem %.2f GB'
      % (a.q, a.t, a.b, a.feat_grad, a.dropout, a.recompute, dt * 1e3,
         '' if dg is None else '   captured as one hipGraph (CapturedTrainStep, camera constants refreshed every step): %.2f ms' % (dg * 1e3),
         di * 1e3, torch.cuda.max_memory_allocated() / 1e9))
