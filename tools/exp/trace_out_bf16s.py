"""Per-phase cycle stamps of the ping-pong out-projection (variant library built with -DSBEV_EXP_TRACE)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebev_amd import _lib, dense
M, N, K, nimg = 900, 256, 32768, 3
x = torch.randn(M, K, device='cuda').clamp_min(0); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
wp = dense.pack_bf16s_frags(w, nimg)
for _ in range(3):
    dense.linear_splitk_bf16s(x, wp, b, nimg=nimg)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 512 * 8))()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.sbev_debug_trace_read.argtypes = [ctypes.c_void_p]
assert raw.sbev_debug_trace_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 512, 8).astype(np.int64)
for grp in (0, 1):
    print('half', grp, ' (cycles) frag reads | W loads | split+stage+X loads | barrier | MFMAs (+W loads) | barrier | -> next')
    for g in list(range(0, 6)) + [14, 15, 28, 29]:
        r = t[grp, g]
        nxt = t[grp, g + 1, 0]
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], nxt - r[6]]
        print('  s=%2d' % g, ' '.join('%6d' % v for v in d), '  total', nxt - r[0])
