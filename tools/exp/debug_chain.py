"""Row-chain kernels (csrc/row_chain.hip) against the op-by-op launches: weight packing vs numpy, output differences per layer
at a small and at the c2 shape, and the step time with / without the chains.  Run on the GPU box."""
import copy
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from sparsebev_amd import _lib, runtime, synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer

DEV = 'cuda:0'
PREFIX = 'decoder.decoder_layer.'


def build(T, L, seed, num_layers):
    params = S.make_params(seed, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=num_layers, num_levels=L, num_classes=10, code_size=10,
                             pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    return m.to(DEV).eval()


def check_pack():
    torch.manual_seed(0)
    m = build(8, 4, 1, 1)
    rt = m.decoder._runtime if hasattr(m.decoder, '_runtime') else None
    ih, iw, sizes = S.PYRAMIDS['tiny'] if 'tiny' in S.PYRAMIDS else S.PYRAMIDS['r50_704x256']
    return m


def run(name, pyr, B, Q, T, layers, seed=5):
    ih, iw, sizes = S.PYRAMIDS[pyr]
    feats = S.make_features(B, T, sizes, seed=seed, device=DEV)
    bbox, feat = S.make_queries(B, Q, seed=seed + 1)
    metas = S.make_img_metas(B, T, ih, iw)
    m = build(T, len(sizes), seed + 2, layers)
    qb, qf = bbox.to(DEV), feat.to(DEV)
    with torch.no_grad():
        runtime.row_chain(True)
        a = m(qb, qf, list(feats), None, copy.deepcopy(metas))
        runtime.row_chain(False)
        b = m(qb, qf, list(feats), None, copy.deepcopy(metas))
        runtime.row_chain(True)
    torch.cuda.synchronize()
    for l in range(layers):
        print('%s layer %d: cls diff %.3e (max |cls| %.2f)  box diff %.3e  finite %s' % (
            name, l, (a[0][l] - b[0][l]).abs().max().item(), b[0][l].abs().max().item(), (a[1][l] - b[1][l]).abs().max().item(),
            bool(torch.isfinite(a[0][l]).all())))
    return m, (qb, qf, feats, metas)


def pack_check(m):
    """the packed image against a numpy restatement of the layout"""
    rt = m.decoder.runtime if hasattr(m.decoder, 'runtime') else None
    for attr in ('_rt', 'runtime', '_runtime'):
        rt = getattr(m.decoder, attr, None) or rt
    if rt is None or rt._keep is None or 'chain_pack' not in rt._keep:
        print('pack: runtime not bound / no chain pack', rt)
        return
    pk = rt._keep['chain_pack'].cpu().numpy()
    W = rt._keep['ffn0_w'].cpu().numpy()          # first block of the image: [512, 256]
    N, K = W.shape
    want = np.zeros(((N + 63) // 64, K // 16, 4, 64, 4), np.float32)
    for cg in range(want.shape[0]):
        rows = W[cg * 64:(cg + 1) * 64]
        want[cg, :, :, :rows.shape[0], :] = rows.reshape(rows.shape[0], K // 16, 4, 4).transpose(1, 2, 0, 3)
    got = pk[:want.size].reshape(want.shape)
    print('pack ffn0: equal', np.array_equal(got, want))


def bench(m, args, n=30):
    qb, qf, feats, metas = args
    from sparsebev_amd.transformer import FeaturePyramid
    pyr = FeaturePyramid(feats)
    for on in (True, False, True, False):
        runtime.row_chain(on)
        with torch.no_grad():
            for _ in range(5):
                m(qb, qf, pyr, None, copy.deepcopy(metas))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                m(qb, qf, pyr, None, copy.deepcopy(metas))
            torch.cuda.synchronize()
        print('row chain %s: %.3f ms / step (eager launches, pyramid resident)' % (on, (time.perf_counter() - t0) / n * 1e3))
    runtime.row_chain(True)


if __name__ == '__main__':
    m, args = run('small', 'tiny' if 'tiny' in S.PYRAMIDS else 'r50_704x256', 2, 49, 2, 3)
    pack_check(m)
    m, args = run('c2', 'r50_704x256', 1, 900, 8, 6)
    bench(m, args)
