"""Dev tool: the on-demand relayout's move kernels against the dense relayout at a bench config's pyramid (HIP events, 20 launches each):
  * every unit marked (the adversarial case: the lazy path must cost what the dense pass costs -- same tile code),
  * the fraction the decoder's layer 0 really marks at that config (tools/relayout_footprint.py), drawn at random,
  * nothing marked (the floor of a launch: flag reads only).
    python tools/bench_lazy_move.py --config c2 [--frac 0.457]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsebev_amd import _lib, dense, synthetic as S

CFG = {'c2': ('r50_704x256', 8, 1, 0.457), 'c3': ('r50_704x256', 8, 8, 0.410), 'c4': ('r101_1408x512', 8, 4, 0.365)}
ap = argparse.ArgumentParser()
ap.add_argument('--config', default='c2', choices=sorted(CFG))
ap.add_argument('--frac', type=float, default=None)
a = ap.parse_args()
pyr, T, B, frac = CFG[a.config]
frac = a.frac if a.frac is not None else frac
ih, iw, sizes = S.PYRAMIDS[pyr]
dev = 'cuda:0'
lib = _lib.load()
n_img, R = B * T * 6, 256
src = [torch.randn(n_img, R, h * w, device=dev) for h, w in sizes]
out = [torch.empty(n_img, h * w, R, device=dev) for h, w in sizes]
hw = [h * w for h, w in sizes]
L = len(hw)
total = int(lib.sbev_lazy_relayout_tiles(L, (ctypes.c_int32 * L)(*hw), n_img, R))
need = torch.zeros(total, 4, device=dev, dtype=torch.uint8)
done = torch.zeros(total, 4, device=dev, dtype=torch.uint8)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
c_hw = (ctypes.c_int32 * L)(*hw)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def lazy():
    _lib.check(lib.sbev_nchw_to_nhwc_lazy(None, None, vp(src), vp(out), L, c_hw, n_img, R, 0, ctypes.c_void_p(need.data_ptr()),
                                          ctypes.c_void_p(done.data_ptr()), 1, 0, st), 'sbev_nchw_to_nhwc_lazy')


def dense_all():
    for s, o in zip(src, out):
        _lib.check(lib.sbev_nchw_to_nhwc_f32(ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(o.data_ptr()), n_img, R, s.shape[2], st), 'sbev_nchw_to_nhwc_f32')


byts = 2 * sum(t.numel() for t in src) * 4
t_dense = timed(dense_all)
need.fill_(1)
t_all = timed(lazy)
ok = all(torch.equal(o, s.permute(0, 2, 1)) for o, s in zip(out, src))
need.copy_((torch.rand(total, 4, device=dev) < frac).to(torch.uint8))
t_frac = timed(lazy)
need.zero_()
t_none = timed(lazy)
print('%s pyramid %s, %d images, %.0f MB in + out: dense relayout (%d launches) %.1f us = %.2f TB/s | lazy, every unit marked %.1f us (%+.1f %%, == dense: %s) | '
      '%.1f %% of the units marked %.1f us = %.2f TB/s | nothing marked %.1f us'
      % (a.config, pyr, n_img, byts / 1e6, L, t_dense, byts / t_dense / 1e6, t_all, 100 * (t_all / t_dense - 1), ok, 100 * frac, t_frac, frac * byts / t_frac / 1e6, t_none))
