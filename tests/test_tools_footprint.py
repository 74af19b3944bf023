"""tools/sampler_footprint.py (the CPU model behind DESIGN_HISTORY.md section 10.8) on a small case: its invariants, and that the orders it
compares are permutations.  CPU only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import sampler_footprint as F                      # noqa: E402


def test_footprint_model_invariants():
    Q, T, P = 64, 2, 4
    pts, bbox, (ih, iw, sizes) = F.layer0_points('r50_704x256', Q, T, P)
    assert pts.shape == (Q, T, F.G, P, 3)
    keys, view, hit = F.taps(pts, ih, iw, sizes, T)
    assert keys.shape == (Q, T, F.G, P, len(sizes), 4)
    req = keys[keys >= 0]
    distinct = np.unique(req).size
    assert 0 < distinct <= req.size < keys.size            # some corners / levels fall outside their maps and request nothing
    assert 0.5 < hit.mean() <= 1.0                         # most points see a camera in the synthetic rig
    orders = F.orders(bbox, view, Q)
    for name, o in orders.items():
        assert sorted(o.tolist()) == list(range(Q)), name
    cap = 1 << 20                                          # a cache that holds everything: misses = distinct segments per XCD
    for mapping in ('launch', 'sector'):
        streams = F.xcd_streams(keys, orders['azimuth'], mapping, inflight=16)
        assert sum(s.size for s in streams) == req.size    # every requested tap appears exactly once
        per_xcd = sum(np.unique(s).size for s in streams)
        assert per_xcd >= distinct
        assert sum(F.lru_misses(s.tolist(), cap) for s in streams) == per_xcd
    # a tiny cache can only do worse, never better than the per-XCD distinct count
    s0 = F.xcd_streams(keys, orders['raster'], 'launch', inflight=16)
    assert sum(F.lru_misses(s.tolist(), 8) for s in s0) >= sum(np.unique(s).size for s in s0)


def test_sector_mapping_gives_every_xcd_one_contiguous_piece_of_the_group_major_list():
    Q = 10
    keys = -np.ones((Q, 2, F.G, 4, 1, 4), dtype=np.int64)
    # tag every (query, group) item with its own key so that the streams reveal which item an XCD got
    for q in range(Q):
        for g in range(F.G):
            keys[q, :, g] = q * F.G + g
    order = np.arange(Q)[::-1].copy()
    streams = F.xcd_streams(keys, order, 'sector', inflight=4)
    got = [sorted(set(s.tolist())) for s in streams]
    per = (Q * F.G + 7) // 8
    for x in range(8):
        want = []
        for m in range(x * per, min((x + 1) * per, Q * F.G)):
            g, pos = divmod(m, Q)
            want.append(int(order[pos]) * F.G + g)
        assert got[x] == sorted(want), x
