"""Dev tool (variant library built with -DSBEV_MIX_TRACE, selected with SBEV_LIB_PATH): wall-clock phase stamps of every workgroup of the
fused gather + mixing launch inside an eager decoder step at config 2 -- where a workgroup's lifetime goes, how many workgroups a CU
holds at a time, and which XCD a block lands on."""
import copy, ctypes, os, sys
import numpy as np
import torch
os.environ.setdefault('SBEV_NO_GRAPH', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                           # noqa: E402
from sparsebev_amd import _lib, runtime, synthetic as S   # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c2'
pyr, Q, T, B, fdtype, P = bench.cfg_fields(bench.CONFIGS[cfgname])
ih, iw, sizes = S.PYRAMIDS[pyr]
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
model = bench.build_model(T, len(sizes), dev, P)
model.decoder.gemm_mode = 'f16x3'
feats = S.make_features(B, T, sizes, seed=0, device=dev, dtype=fdtype)
feats = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]
bbox, qfeat = [t.to(dev) for t in S.make_queries(B, Q, seed=0)]
metas = S.make_img_metas(B, T, ih, iw)
for _ in range(4):
    model(bbox, qfeat, list(feats), None, copy.deepcopy(metas))
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
SLOTS, n = 16, B * Q * 4
buf = (ctypes.c_longlong * (SLOTS * n))()
raw.sbev_debug_mix_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert raw.sbev_debug_mix_trace_read(buf, SLOTS * n) == 0
t = np.array(buf, dtype=np.int64).reshape(n, SLOTS)
t0 = t[:, 0].min()
names = ['gather (wave 0: 2 units)', 'wait for the other waves', 'x fragments from LDS', 'matmul 1 (+ M arrives)', 'LN1 stats + S to LDS (S arrives)',
         'LN1 exchange', 'matmul 2', 'LN2 + transpose', 'stores issued']
d = (t[:, 1:10] - t[:, 0:9]) * 0.01           # us
life = (t[:, 9] - t[:, 0]) * 0.01
print('%s: %d workgroups; kernel span (first start -> last end) %.1f us; workgroup lifetime mean %.2f median %.2f p10 %.2f p90 %.2f us' % (
    cfgname, n, (t[:, 9].max() - t0) * 0.01, life.mean(), np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
for i, nm in enumerate(names):
    print('  %-36s mean %6.2f  median %6.2f  p90 %6.2f us' % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
if t[:, 10].min() > 0:
    g = (t[:, 11] - t[:, 10]) * 0.01; w = (t[:, 12] - t[:, 11]) * 0.01; r = (t[:, 1] - t[:, 12]) * 0.01; f = (t[:, 10] - t[:, 0]) * 0.01
    print('  wave 0: first unit incl. its loc / weight fetch %.2f us; second unit: geometry + 16 tap requests %.2f, wait + consume %.2f, reduce + LDS (to the end of the gather) %.2f us' % (f.mean(), g.mean(), w.mean(), r.mean()))
if t[:, 13].min() > 0:
    print('    second unit in detail: slab bases + phase 1 (4 shuffles + geometry) %.2f us, phase 2 (32 ds_bpermute) %.2f us, 16 tap requests issued %.2f us' % (
        ((t[:, 13] - t[:, 10]) * 0.01).mean(), ((t[:, 14] - t[:, 13]) * 0.01).mean(), ((t[:, 11] - t[:, 14]) * 0.01).mean()))
xcc = (t[:, 15] >> 32) & 0xf
blk = np.arange(n)
print('block %% 8 == XCC_ID for %.1f %% of the blocks; (XCC_ID - block) %% 8 histogram: %s' % (100.0 * (xcc == blk % 8).mean(), np.bincount((xcc - blk) % 8, minlength=8).tolist()))
cu = t[:, 15] & 0xffffffff
print('distinct (XCC, HW_ID[cu/sh/se bits 8..15]) pairs: %d' % len(set(zip(xcc.tolist(), ((cu >> 8) & 0xff).tolist()))))
start = (t[:, 0] - t0) * 0.01
end = (t[:, 9] - t0) * 0.01
for us in (5, 15, 25, 35, 45, 55, 65, 75):
    print('  t = %2d us: %4d workgroups resident (%.2f per CU)' % (us, int(((start <= us) & (end > us)).sum()), ((start <= us) & (end > us)).sum() / 256.0))
# by launch round: workgroups that started in the first 3 us (round 1) vs later
first = start < 3.0
print('first-round workgroups: %d, lifetime mean %.2f us; later ones: lifetime mean %.2f us' % (first.sum(), life[first].mean(), life[~first].mean()))
for i, nm in enumerate(names):
    print('    %-36s first %6.2f  later %6.2f' % (nm, d[first, i].mean(), d[~first, i].mean()))
