#!/usr/bin/env python
"""Footprint model of the adaptive-sampling gather (CPU only, numpy): which 256-byte tap segments does one decoder layer
request, how many of them are distinct, and how many reach the fabric under a per-XCD LRU model of the 4-MiB L2 for a given
ORDER of the (query, group) items over the 8 XCDs.

Why: rocprofv3 shows the gather's L2 hit ratio at 0.25-0.38 and ~190 MB of fabric reads per launch at config 2 against 472 MB
of requested tap bytes.  Before building a per-layer ordering kernel this answers, without a GPU, (1) what
the floor is -- distinct segments, chip-wide and summed over the XCDs that touch them (the L2s are private) -- and (2) what an
ordering can recover: the fused kernel's launch order (block b = item (q, g) = (b / 4, b % 4) on XCD b % 8) against a random
query order and against azimuth-sector orders in which an XCD owns ONE group and one half of the camera ring.

The model (deliberately simple; compare the `launch` row with profiles/r4_pmc_<config>.json):
  * geometry exactly as the decoder's layer 0 on the bench inputs (synthetic.make_queries / camera_rig / the model's
    sampling_offset Linear): points -> first hitting camera -> per level the 4 bilinear corners, a corner outside the map or a
    level outside (-1, H) x (-1, W) requests nothing;
  * a tap = one 256-byte segment (64 channels of a group, fp32; 128 bytes with bf16 storage) of slab (frame, group), keyed by
    (frame, group, level, view, y, x);
  * per XCD an LRU set of 4 MiB / segment bytes; the XCD walks its blocks in launch order with `inflight` workgroups
    interleaved (wave w of a workgroup gathers frames w, w + 4: two rounds of 4 frames), every miss is one fabric read.
No feature data is touched; the oracle is not used (this is a design tool, not a checker).

    python tools/sampler_footprint.py --config c2 [--json profiles/r4_sampler_footprint_c2.json]
"""
import argparse
import json
import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebev_amd import synthetic as S                      # noqa: E402

CONFIGS = {   # pyramid, Q, T, P, bytes per channel (bench.py CONFIGS)
    'c2': ('r50_704x256', 900, 8, 4, 4),
    'c5': ('eva02_1600x640', 900, 8, 4, 2),
    'c6': ('eva02_1600x640', 1600, 15, 8, 2),
}
G, N_VIEWS, C = 4, 6, 64


def layer0_points(pyr, Q, T, P, seed=0):
    """sample points of layer 0 on the bench inputs: [Q, T, G, P, 3] in metres (models/sparsebev_transformer.py:270-300)."""
    from sparsebev_amd.transformer import SparseBEVTransformer
    ih, iw, sizes = S.PYRAMIDS[pyr]
    torch.manual_seed(0)
    m = SparseBEVTransformer(256, num_frames=T, num_points=P, num_layers=6, num_levels=len(sizes), num_classes=10, code_size=10,
                             pc_range=S.PC_RANGE)
    m.init_weights()
    S.randomize_zero_init(m, std=0.02, seed=0)
    sd = m.state_dict()
    w = sd['decoder.decoder_layer.sampling.sampling_offset.weight'].double()
    b = sd['decoder.decoder_layer.sampling.sampling_offset.bias'].double()
    bbox, feat = S.make_queries(1, Q, seed=seed)
    bbox, feat = bbox[0].double(), feat[0].double()
    off = (feat @ w.t() + b).reshape(Q, G * P, 3)
    pc = torch.tensor(S.PC_RANGE, dtype=torch.float64)
    xyz = bbox[:, 0:3] * (pc[3:6] - pc[0:3]) + pc[0:3]
    wlh = bbox[:, 3:6].exp()
    ang = torch.atan2(bbox[:, 6], bbox[:, 7])
    d = off * wlh[:, None, :]
    c, s = ang.cos()[:, None], ang.sin()[:, None]
    dx, dy = d[..., 0] * c - d[..., 1] * s, d[..., 0] * s + d[..., 1] * c
    pts = torch.stack([xyz[:, None, 0] + dx, xyz[:, None, 1] + dy, xyz[:, None, 2] + d[..., 2]], dim=-1)     # [Q, GP, 3]
    tdiff = 0.5 * torch.arange(T, dtype=torch.float64)                                                      # ts[0] - ts[f]
    pts = pts[:, None].expand(Q, T, G * P, 3).clone()
    pts[..., 0:2] -= bbox[:, None, None, 8:10] * tdiff[None, :, None, None]
    return pts.reshape(Q, T, G, P, 3).numpy(), bbox.numpy(), (ih, iw, sizes)


def taps(pts, ih, iw, sizes, T):
    """int64 keys [Q, T, G, P, L, 4] of the requested segments (-1: nothing requested), key = (((t*G+g)*L+l)*N+view)*HWmax + y*W+x."""
    Q, _, _, P, _ = pts.shape
    rig = S.camera_rig(T, ih, iw).reshape(T, N_VIEWS, 4, 4)
    ph = np.concatenate([pts, np.ones(pts.shape[:-1] + (1,))], axis=-1)                   # [Q,T,G,P,4]
    cam = np.einsum('tnij,qtgpj->qtgpni', rig, ph)                                        # [Q,T,G,P,N,4]
    homo = cam[..., 2]
    hn = np.maximum(homo, 1e-5)
    u, v = cam[..., 0] / hn / iw, cam[..., 1] / hn / ih
    valid = (homo > 1e-5) & (u > 0) & (u < 1) & (v > 0) & (v < 1)
    view = np.argmax(valid, axis=-1)                                                      # first hit, 0 when none
    uu = np.take_along_axis(u, view[..., None], -1)[..., 0]
    vv = np.take_along_axis(v, view[..., None], -1)[..., 0]
    L = len(sizes)
    hwmax = max(h * w for h, w in sizes)
    keys = np.full(pts.shape[:-1] + (L, 4), -1, dtype=np.int64)
    tt = np.arange(T)[None, :, None, None]
    gg = np.arange(G)[None, None, :, None]
    for l, (H, W) in enumerate(sizes):
        x, y = uu * (W - 1), vv * (H - 1)
        lvl_ok = (y > -1) & (y < H) & (x > -1) & (x < W)
        x0, y0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
        for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            xi, yi = x0 + dx, y0 + dy
            ok = lvl_ok & (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            key = ((((tt * G + gg) * L + l) * N_VIEWS + view) * hwmax) + yi * W + xi
            keys[..., l, k] = np.where(ok, key, -1)
    return keys, view, valid.any(-1)


def lru_misses(stream, capacity):
    cache = OrderedDict()
    miss = 0
    for k in stream:
        if k in cache:
            cache.move_to_end(k)
        else:
            miss += 1
            cache[k] = None
            if len(cache) > capacity:
                cache.popitem(last=False)
    return miss


def xcd_streams(keys, order, mapping, inflight):
    """access streams of the 8 XCDs.  keys [Q,T,G,P,L,4]; order: permutation of the queries (position -> query).
    mapping 'launch': block b = (position b / G, group b % G) on XCD b % 8 (the fused kernel today);
    mapping 'sector': XCD x owns group x / 2 and positions [half * Q/2, ...) with half = x % 2, walked in order."""
    Q, T = keys.shape[0], keys.shape[1]
    items = [[] for _ in range(8)]
    if mapping == 'launch':
        for b in range(Q * G):
            items[b % 8].append((order[b // G], b % G))
    else:
        per = (Q * G + 7) // 8
        for m in range(Q * G):                       # group-major list, XCD x takes the x-th eighth
            g, pos = divmod(m, Q)
            items[m // per].append((order[pos], g))
    rounds = [list(range(r, T, 4)) for r in range(4)]       # wave w gathers frames w, w + 4, ...
    nr = max(len(r) for r in rounds)
    streams = []
    for x in range(8):
        out = []
        it = items[x]
        for w0 in range(0, len(it), inflight):
            win = it[w0:w0 + inflight]
            for step in range(nr):
                for (q, g) in win:
                    for w in range(4):
                        if step < len(rounds[w]):
                            k = keys[q, rounds[w][step], g].reshape(-1)
                            out.append(k[k >= 0])
        streams.append(np.concatenate(out) if out else np.zeros(0, np.int64))
    return streams


def orders(bbox, view0, Q, seed=1):
    """query orders: the launch order (head init = BEV raster), a random shuffle, azimuth (angle of the box centre around the ego),
    and azimuth within range rings."""
    pc = S.PC_RANGE
    x = bbox[:, 0] * (pc[3] - pc[0]) + pc[0]
    y = bbox[:, 1] * (pc[4] - pc[1]) + pc[1]
    az = np.arctan2(y, x)
    r = np.hypot(x, y)
    rng = np.random.default_rng(seed)
    ring = np.minimum((r / 12.0).astype(np.int64), 4)
    snake = np.where(ring % 2 == 0, az, -az)
    return {
        'raster': np.arange(Q),
        'shuffle': rng.permutation(Q),
        'azimuth': np.argsort(az, kind='stable'),
        'ring_azimuth': np.lexsort((snake, ring)),
        'azimuth_range': np.lexsort((r, (az / (2 * math.pi / 48)).astype(np.int64))),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--inflight', type=int, default=112, help='workgroups an XCD interleaves (32 CUs x 3-4 resident)')
    ap.add_argument('--l2-mib', type=float, default=4.0)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    pyr, Q, T, P, bpc = CONFIGS[args.config]
    seg = C * bpc
    pts, bbox, (ih, iw, sizes) = layer0_points(pyr, Q, T, P)
    keys, view, hit = taps(pts, ih, iw, sizes, T)
    L = len(sizes)
    req = int((keys >= 0).sum())
    distinct = int(np.unique(keys[keys >= 0]).size)
    per_level = []
    for l in range(L):
        kl = keys[..., l, :]
        per_level.append({'level': l, 'size': list(sizes[l]), 'requested': int((kl >= 0).sum()), 'distinct': int(np.unique(kl[kl >= 0]).size),
                          'map_segments': int(sizes[l][0] * sizes[l][1] * N_VIEWS * T * G)})
    cap = int(args.l2_mib * 1024 * 1024 / seg)
    res = {'config': args.config, 'segment_bytes': seg, 'taps_issued': int(keys.size), 'requested_segments': req,
           'requested_MB': round(req * seg / 1e6, 1), 'distinct_segments': distinct, 'distinct_MB': round(distinct * seg / 1e6, 1),
           'points_without_camera': round(float(1.0 - hit.mean()), 4), 'per_level': per_level, 'l2_segments_per_xcd': cap,
           'inflight': args.inflight, 'orders': {}}
    print('%s: %d tap slots, %d requested (%.1f MB), %d distinct (%.1f MB = the chip-wide floor)' %
          (args.config, keys.size, req, req * seg / 1e6, distinct, distinct * seg / 1e6))
    for pl in per_level:
        print('  level %d %3dx%-3d requested %8d distinct %8d (%.2f)  map %8d' % (pl['level'], pl['size'][0], pl['size'][1], pl['requested'],
                                                                                 pl['distinct'], pl['distinct'] / max(1, pl['requested']), pl['map_segments']))
    for name, order in orders(bbox, view, Q).items():
        for mapping in ('launch', 'sector'):
            if mapping == 'sector' and name in ('raster', 'shuffle'):
                continue
            streams = xcd_streams(keys, order, mapping, args.inflight)
            xcd_distinct = sum(int(np.unique(s).size) for s in streams)
            miss = sum(lru_misses(s.tolist(), cap) for s in streams)
            tot = sum(s.size for s in streams)
            res['orders']['%s/%s' % (name, mapping)] = {'xcd_distinct_MB': round(xcd_distinct * seg / 1e6, 1), 'lru_fabric_MB': round(miss * seg / 1e6, 1),
                                                        'l2_hit': round(1.0 - miss / max(1, tot), 4)}
            print('  %-14s %-7s distinct summed over XCDs %7.1f MB   LRU fabric reads %7.1f MB   L2 hit %.3f' %
                  (name, mapping, xcd_distinct * seg / 1e6, miss * seg / 1e6, 1.0 - miss / max(1, tot)))
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
