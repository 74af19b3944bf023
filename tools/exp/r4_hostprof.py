"""Dev tool: where the host time of a replayed step goes (fresh tensors every step)."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from sparsebev_amd import synthetic as S
from test_gpu_stepgraph import build
DEV = "cuda:0"
torch.set_grad_enabled(False)
B, Q, T = 1, 900, 8
ih, iw, sizes = S.PYRAMIDS['r50_704x256']
g = build(T, len(sizes), 16, num_layers=6)
metas = S.make_img_metas(B, T, ih, iw)
base = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=41)]
bbox0, feat0 = [t.to(DEV) for t in S.make_queries(B, Q, seed=42)]
for _ in range(4):
    g(bbox0.clone(), feat0.clone(), [f.clone() for f in base], None, metas)
torch.cuda.synchronize()
for rep in range(3):
    sets = [([f.clone() for f in base], bbox0.clone(), feat0.clone()) for _ in range(6)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for feats, bbox, feat in sets:
        g(bbox, feat, feats, None, metas)
    host = (time.perf_counter() - t0) / 6
    torch.cuda.synchronize()
    print('host issue per replayed step: %.3f ms' % (host * 1e3))
sets = [([f.clone() for f in base], bbox0.clone(), feat0.clone()) for _ in range(6)]
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for feats, bbox, feat in sets:
    g(bbox, feat, feats, None, metas)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])
