"""Dev tool (variant library built with -DSBEV_SASA_TRACE): phase stamps of wave 0 of workgroup 0 of sasa_kernel at c2."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebev_amd import _lib, dense
lib = _lib.load()
B, Q, H = 1, 900, 8
g = torch.Generator().manual_seed(1)
qkvt = torch.randn(B, Q, 776, generator=g).cuda()
bbox = torch.rand(B, Q, 10, generator=g).cuda()
out = torch.empty(B, Q, 256, device='cuda')
pc = (ctypes.c_double * 6)(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
raw = ctypes.CDLL(_lib.LIB_PATH)
for _ in range(5):
    assert raw.sbev_sasa_f32(p(qkvt), ctypes.c_int64(776), p(bbox), pc, None, p(out), B, Q, H, 32, st) == 0
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
raw.sbev_debug_sasa_trace_read.argtypes = [ctypes.c_void_p]
assert raw.sbev_debug_sasa_trace_read(buf) == 0
t = np.array(buf, dtype=np.int64)
print('prologue (Q frags, centres, first K/V tiles -> LDS, barrier): %d cycles' % (t[1] - t[0]))
prev = t[1]
for it in range(4):
    b = 2 + 8 * it
    if t[b] == 0:
        break
    print('  iteration %d: early tile I/O %d | S^T = K Q^T %d | bias + softmax %d | PV %d | late tile I/O %d | barrier %d   total %d' % (
        it, t[b] - prev, t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], t[b + 4] - t[b + 3], t[b + 5] - t[b + 4], t[b + 6] - prev))
    prev = t[b + 6]
print('merge + store: %d cycles; kernel (this wave): %d cycles' % (t[63] - prev, t[63] - t[0]))
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for s_, e_ in ev:
    s_.record(); raw.sbev_sasa_f32(p(qkvt), ctypes.c_int64(776), p(bbox), pc, None, p(out), B, Q, H, 32, st); e_.record()
torch.cuda.synchronize()
ts = sorted(s_.elapsed_time(e_) for s_, e_ in ev)
print('HIP-event median per launch: %.1f us' % (ts[len(ts) // 2] * 1e3))
