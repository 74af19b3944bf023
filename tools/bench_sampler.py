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
    ap.add_argument('--uniform', action='store_true', help='replace the projected sample locations by uniform random ones (no clustering)')
    ap.add_argument('--bwd', action='store_true', help='time sbev_msmv_bwd (the backward kernel alone, grad buffers pre-zeroed once)')
    a = ap.parse_args()
    from test_gpu_sampling import c2_inputs
    import test_gpu_sampling as tg
    ih, iw, sizes = S.PYRAMIDS[a.pyramid]
    feats, pts, l2i, loc, wbp, _, (ih, iw, B, Q, T, G, P, L) = c2_inputs(a.B, a.Q, a.T) if a.pyramid == 'r50_704x256' else (None,) * 7
    if a.bf16:
        feats = [f.to(torch.bfloat16) for f in feats]
    if a.uniform:
        loc = torch.rand_like(loc)
        loc[..., 2] = torch.randint(0, 6, loc.shape[:-1], device=loc.device).float() / 5
    layout = ops.OUT_REF if a.layout == 'ref' else ops.OUT_MIX
    if a.bwd:
        import ctypes
        from sparsebev_amd import _lib
        lib = _lib.load()
        Ln = len(feats)
        Bp, N, _, _, C = feats[0].shape
        _, Qn, Pn, _ = loc.shape
        gout = torch.randn(Bp, Qn, C, Pn, device=loc.device)
        gfeats = [torch.zeros_like(f) for f in feats]
        gloc, gw = torch.empty_like(loc), torch.empty_like(wbp)
        hw = [(f.shape[2], f.shape[3]) for f in feats]
        c_feats = (ctypes.c_void_p * Ln)(*[f.data_ptr() for f in feats])
        c_gfeats = (ctypes.c_void_p * Ln)(*[f.data_ptr() for f in gfeats])
        c_hw = (ctypes.c_int32 * (2 * Ln))(*[v for pair in hw for v in pair])
        c_sbo = (ctypes.c_int64 * Ln)(*[N * h * w * C for h, w in hw])
        c_sv = (ctypes.c_int64 * Ln)(*[h * w * C for h, w in hw])
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        fn = lambda: lib.sbev_msmv_bwd(c_feats, c_gfeats, c_hw, Ln, Bp, N, C, Qn, Pn, 1, c_sbo, 0, c_sv, C, p(loc), p(wbp), p(gout), p(gloc), p(gw), st)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
        for s_, e_ in evs:
            s_.record(); fn(); e_.record()
        torch.cuda.synchronize()
        ts = sorted(s_.elapsed_time(e_) for s_, e_ in evs)
        npts = loc.shape[0] * loc.shape[1] * loc.shape[2]
        # per point: L taps x 4 corners x C floats read AND atomically added, C grad_out floats, coords / weights in, their grads out
        bytes_ = npts * (Ln * 4 * C * 4 * 2 + C * 4 + 12 + 4 * Ln + 12 + 4 * Ln)
        print('BWD points %d  bytes %.1f MB  median %.1f us  min %.1f us  -> %.0f GB/s algorithmic; %.2f G atomic dwords/s'
              % (npts, bytes_ / 1e6, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, bytes_ / ts[len(ts) // 2] / 1e6, npts * Ln * 4 * C / ts[len(ts) // 2] / 1e6))
        return
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
