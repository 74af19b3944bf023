"""Eager runtime call vs hipGraph replay of the same decoder step (features resident channels-last)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsebev_amd import synthetic as S
from sparsebev_amd.runtime import DecoderRuntime
from sparsebev_amd.transformer import SparseBEVTransformer, FeaturePyramid, DecoderContext

dev = torch.device('cuda:0')
T, L, Q, B = 8, 4, 900, 1
ih, iw, sizes = S.PYRAMIDS['r50_704x256']
params = S.make_params(0, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=6, num_levels=L, num_classes=10, code_size=10, pc_range=S.PC_RANGE)
m.load_state_dict({'decoder.decoder_layer.' + k: v for k, v in params.items()})
m = m.to(dev).eval()
feats = S.make_features(B, T, sizes, seed=0, device=dev)
pyr, ctx = FeaturePyramid(feats), DecoderContext(S.make_img_metas(B, T, ih, iw), B, dev)
bbox, feat = [t.to(dev) for t in S.make_queries(B, Q, seed=0)]
eager = DecoderRuntime(m.decoder)
graph = DecoderRuntime(m.decoder).capture(bbox, feat, pyr, ctx)


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    print('eager %.4f ms   graph %.4f ms (%d nodes)' % (timeit(lambda: eager.forward(bbox, feat, pyr, ctx)), timeit(graph.replay), graph.num_nodes))
