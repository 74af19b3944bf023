#!/usr/bin/env python
"""Which part of the feature pyramid does the decoder READ?  (CPU only; design tool for the on-demand relayout)

The reference regroups EVERY pixel of every level on every call (models/sparsebev_transformer.py:73-85) and so did this repo's
NCHW -> NHWC relayout (csrc/layout.hip::transpose_tiles_kernel).  The gather only reads the 4 bilinear corners of each sample
point at each level (models/csrc/msmv_sampling/msmv_sampling_forward.cu:27-66, sparsebev_sampling.py:88-109).  This tool runs
the pinned CPU oracle's 6 free-running decoder layers on the bench inputs (synthetic seed 0), takes the recorded `loc_bp` taps
of every layer and counts the relayout units a tap touches, at two granularities:

  * tile   = 64 consecutive pixels of one image of one level x all 256 channels (64 KB fp32);
  * unit   = the same 64 pixels x ONE 64-channel group (16 KB fp32: what one workgroup of the relayout kernel moves; the groups
             have their own sample points, so group g only needs ITS channel slice of the pixels it touches).

Printed / stored per config: the touched fraction (by bytes) after layer 0 and after all six layers, per level, and the bytes the
layers after the first one add.

    python tools/relayout_footprint.py --config c2 [--json profiles/r6_relayout_footprint_c2.json]

The oracle is used here as a design tool (tools/ is not the product path).  c3 / c4 run one sample of the batch at a time.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebev_amd import synthetic as S                      # noqa: E402

CONFIGS = {   # pyramid, Q, T, per-GPU batch, sampling points (bench.py CONFIGS)
    'c1': ('r50_704x256', 100, 1, 1, 4),
    'c2': ('r50_704x256', 900, 8, 1, 4),
    'c3': ('r50_704x256', 400, 8, 8, 4),
    'c4': ('r101_1408x512', 900, 8, 4, 4),
}
G, N_VIEWS, TILE = 4, 6, 64


def touched_units(loc_bp, sizes, T, Q, P):
    """loc_bp [T*G, Q, P, 3] of ONE sample -> per level a bool array [T*N_VIEWS, tiles_l, G]: the sampler's own corner rule
    (csrc/msmv_chunk.inc phase 1 == msmv_sampling_forward.cu:41-66): level skipped unless -1 < h_im < H and -1 < w_im < W, a
    corner outside the map reads nothing."""
    loc = loc_bp.reshape(T, G, Q * P, 3).numpy().astype(np.float32)
    out = []
    view = np.clip(np.rint(loc[..., 2] * np.float32(N_VIEWS - 1)).astype(np.int64), 0, N_VIEWS - 1)      # [T,G,QP]
    t_idx = np.arange(T)[:, None, None]
    g_idx = np.broadcast_to(np.arange(G)[None, :, None], view.shape)
    img = t_idx * N_VIEWS + view
    for (H, W) in sizes:
        tiles = (H * W + TILE - 1) // TILE
        need = np.zeros((T * N_VIEWS, tiles, G), dtype=bool)
        h_im = loc[..., 1] * np.float32(H - 1)
        w_im = loc[..., 0] * np.float32(W - 1)
        ok = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        hf = np.clip(np.floor(h_im), -1, H).astype(np.int64)
        wf = np.clip(np.floor(w_im), -1, W).astype(np.int64)
        for dh in (0, 1):
            for dw in (0, 1):
                hc, wc = hf + dh, wf + dw
                inb = ok & (hc >= 0) & (hc <= H - 1) & (wc >= 0) & (wc <= W - 1)
                pix = np.clip(hc, 0, H - 1) * W + np.clip(wc, 0, W - 1)
                need[img[inb], pix[inb] // TILE, g_idx[inb]] = True
        out.append(need)
    return out


def run(config, seed=0, layers=6):
    from oracle import sparsebev_oracle as O
    from sparsebev_amd.transformer import SparseBEVTransformer
    pyr, Q, T, B, P = CONFIGS[config]
    ih, iw, sizes = S.PYRAMIDS[pyr]
    torch.manual_seed(0)
    m = SparseBEVTransformer(256, num_frames=T, num_points=P, num_layers=layers, num_levels=len(sizes), num_classes=10, code_size=10,
                             pc_range=S.PC_RANGE)
    m.init_weights()
    S.randomize_zero_init(m, std=0.02, seed=0)
    params = O.strip_prefix({k: v.detach().float() for k, v in m.state_dict().items()})
    bbox_all, feat_all = S.make_queries(B, Q, seed=seed)
    metas_all = S.make_img_metas(B, T, ih, iw)
    px_per_level = [h * w for h, w in sizes]
    tiles_per_level = [(p + TILE - 1) // TILE for p in px_per_level]
    # accumulated over the samples of the batch: [layer][level] counts
    tile_cnt = np.zeros((layers, len(sizes)), dtype=np.int64)        # tiles (any group) touched after layers 0..l
    unit_cnt = np.zeros((layers, len(sizes)), dtype=np.int64)        # (tile, group) units touched after layers 0..l
    torch.set_num_threads(os.cpu_count() or 1)
    for b in range(B):
        # i.i.d. noise features, one sample at a time (bench.py draws the batch on the device generator: other values, same statistics --
        # the features only steer the boxes of layers 1..5)
        feats = S.make_features(1, T, sizes, seed=seed + b)
        taps = []
        with torch.no_grad():
            O.decoder(params, bbox_all[b:b + 1], feat_all[b:b + 1], feats, metas_all[b:b + 1], S.PC_RANGE, num_layers=layers,
                      num_points=P, taps=taps)
        del feats
        acc = None
        for l, t in enumerate(taps):
            need = touched_units(t['loc_bp'], sizes, T, Q, P)
            acc = need if acc is None else [a | n for a, n in zip(acc, need)]
            for lv, a in enumerate(acc):
                tile_cnt[l, lv] += int(a.any(axis=2).sum())
                unit_cnt[l, lv] += int(a.sum())
    n_img = B * T * N_VIEWS
    bytes_tile = TILE * 256 * 4
    tot_tiles = np.array([n_img * t for t in tiles_per_level])
    res = {'config': config, 'pyramid': pyr, 'Q': Q, 'T': T, 'B': B, 'P': P, 'seed': seed, 'levels': [list(s) for s in sizes],
           'tile': '64 pixels x 256 channels (64 KB fp32)', 'unit': '64 pixels x 64 channels of one group (16 KB fp32)',
           'total_feature_MB': round(float(tot_tiles.sum()) * bytes_tile / 1e6, 1),
           'what': 'pinned CPU oracle, %d free-running layers on synthetic seed %d; corner rule of msmv_sampling_forward.cu:41-66' % (layers, seed)}
    for name, cnt, div in (('tiles', tile_cnt, 1), ('units', unit_cnt, G)):
        frac = cnt.sum(axis=1) / (tot_tiles.sum() * div)
        res[name] = {
            'touched_fraction_layer0': round(float(frac[0]), 4),
            'touched_fraction_all_layers': round(float(frac[-1]), 4),
            'touched_fraction_after_layer': [round(float(f), 4) for f in frac],
            'per_level_fraction_layer0': [round(float(cnt[0, lv]) / (tot_tiles[lv] * div), 4) for lv in range(len(sizes))],
            'per_level_fraction_all_layers': [round(float(cnt[-1, lv]) / (tot_tiles[lv] * div), 4) for lv in range(len(sizes))],
            'MB_added_by_layer': [round(float((cnt[l].sum() - (cnt[l - 1].sum() if l else 0)) * bytes_tile / div) / 1e6, 2) for l in range(layers)],
        }
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--layers', type=int, default=6)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    res = run(args.config, args.seed, args.layers)
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.json:
        with open(args.json, 'w') as f:
            f.write(txt + '\n')


if __name__ == '__main__':
    main()
