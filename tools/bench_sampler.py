"""Micro-benchmark of the sampling kernel alone at a BASELINE config (default c2): algorithmic GB/s
(SURVEY.md section 8d byte model) from HIP-event timing on the launch stream.  Dev tool; bench.py is the
contract."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebev_amd import ops, synthetic as S   # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pyramid', default='r50_704x256')
    ap.add_argument('--B', type=int, default=1)
    ap.add_argument('--Q', type=int, default=900)
    ap.add_argument('--T', type=int, default=8)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--bf16', action='store_true')
    ap.add_argument('--layout', default='ref', choices=['ref', 'mix'])
    a = ap.parse_args()
    from test_gpu_sampling import c2_inputs
    import test_gpu_sampling as tg
    ih, iw, sizes = S.PYRAMIDS[a.pyramid]
    feats, pts, l2i, loc, wbp, _, (ih, iw, B, Q, T, G, P, L) = c2_inputs(a.B, a.Q, a.T) if a.pyramid == 'r50_704x256' else (None,) * 7
    if a.bf16:
        feats = [f.to(torch.bfloat16) for f in feats]
    layout = ops.OUT_REF if a.layout == 'ref' else ops.OUT_MIX
    for _ in range(5):
        ops.msmv_sampling(feats, loc, wbp, out_layout=layout, T=T, G=G)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
    for s, e in evs:
        s.record()
        ops.msmv_sampling(feats, loc, wbp, out_layout=layout, T=T, G=G)
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    npts = loc.shape[0] * loc.shape[1] * loc.shape[2]
    sf = 2 if a.bf16 else 4
    bytes_ = npts * (L * 4 * 64 * sf + 12 + 4 * L + 64 * 4)
    med = ts[len(ts) // 2]
    print('points %d  bytes %.1f MB  median %.1f us  min %.1f us  -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)'
          % (npts, bytes_ / 1e6, med * 1e3, ts[0] * 1e3, bytes_ / med / 1e6, bytes_ / med / 1e6 / 80))


if __name__ == '__main__':
    main()
