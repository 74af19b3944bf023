"""Experiment: one training step (forward + backward, 6 layers, c2) captured as ONE hipGraph (torch.cuda.graph) vs eager."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sparsebev_amd import synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer
dev = 'cuda:0'
ih, iw, sizes = S.PYRAMIDS['r50_704x256']
m = SparseBEVTransformer(256, num_frames=8, num_points=4, num_layers=6, num_levels=len(sizes), pc_range=S.PC_RANGE)
m.init_weights(); S.randomize_zero_init(m, std=0.02, seed=0)
m = m.to(dev).train()
m.decoder.decoder_layer.self_attn.attn_drop = 0.0
m.decoder.decoder_layer.ffn_drop = 0.0
feats = [f for f in S.make_features(1, 8, sizes, seed=0, device=dev)]
bbox, feat = [t.to(dev) for t in S.make_queries(1, 900, seed=0)]
feat.requires_grad_(True)
metas = S.make_img_metas(1, 8, ih, iw)
params = [p for p in m.parameters()]
from sparsebev_amd.transformer import FeaturePyramid, DecoderContext
ctx = DecoderContext(metas, 1, torch.device(dev))      # host -> device copies of the camera matrices / time stamps: outside the capture
def step():
    pyr = FeaturePyramid(list(feats))                  # the NCHW -> NHWC relayout stays inside (device work only)
    cls, box = m.decoder.forward_differentiable(bbox, feat, list(feats), pyr, None, ctx)
    (cls.sum() + box.sum()).backward()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def eager():
    for p in params: p.grad = None
    feat.grad = None
    step()
# everything on ONE side stream from the first step on: the parameters' AccumulateGrad nodes remember the stream they were created
# on, and a backward under capture that has to hop to another (non-capturing) stream ends the capture with a crash inside HIP
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    print('eager train step %.2f ms' % timeit(eager), flush=True)
    for _ in range(3):
        for p in params: p.grad = None
        feat.grad = None
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
for p in params: p.grad = None
feat.grad = None
try:
    with torch.cuda.graph(g, stream=s):
        step()
    gbuf = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}      # the graph's own gradient buffers (static memory)
    gfeat = feat.grad
    print('graph replay train step %.2f ms' % timeit(g.replay), flush=True)
    g.replay(); torch.cuda.synchronize()
    got = {n: t.clone() for n, t in gbuf.items()}
    got_feat = gfeat.clone()
    eager(); torch.cuda.synchronize()
    worst, scale = 0.0, 0.0
    for n, p in m.named_parameters():
        if p.grad is not None and n in got:
            worst = max(worst, (p.grad - got[n]).abs().max().item())
            scale = max(scale, p.grad.abs().max().item())
    print('max |grad(graph) - grad(eager)| = %.3e (largest gradient entry %.3e) over %d parameters; query_feat grad diff %.3e'
          % (worst, scale, len(got), (feat.grad - got_feat).abs().max().item()))
except Exception as e:
    print('capture failed:', repr(e)[:500])
