"""Dev tool: time adaptive_mixing_kernel alone at a decoder shape (default config 2: 900 queries, 4 groups, 32 in-points)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebev_amd import _lib   # noqa: E402

BQ, G, Pin, C, Pout = int(os.environ.get('BQ', 900)), 4, int(os.environ.get('PIN', 32)), 64, 128
lib = _lib.load()
x = torch.randn(BQ, G, Pin, C, device='cuda')
params = torch.randn(BQ, G * (C * C + Pout * Pin), device='cuda') * 0.1
out = torch.empty(BQ, G * Pout * C, device='cuda')
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fn = lambda: lib.sbev_adaptive_mixing_f32(p(x), p(params), p(out), BQ, G, Pin, C, Pout, 1e-5, st)
for _ in range(5):
    fn()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for s, e in evs:
    s.record(); fn(); e.record()
torch.cuda.synchronize()
ts = sorted(s.elapsed_time(e) for s, e in evs)
byt = (x.numel() + params.numel() + out.numel()) * 4
print('mixing BQ=%d Pin=%d: median %.1f us  min %.1f us  %.0f GB/s (%.1f MB)' % (BQ, Pin, ts[25] * 1e3, ts[0] * 1e3, byt / ts[25] / 1e6, byt / 1e6))
