"""Dev tool: time sasa_kernel alone at the decoder's shape (B*Q = 900 queries, 8 heads)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebev_amd import _lib, synthetic as S   # noqa: E402
B, Q, H = int(os.environ.get('B', 1)), int(os.environ.get('Q', 900)), 8
lib = _lib.load()
qkvt = torch.randn(B * Q, 776, device='cuda')
bbox = torch.rand(B, Q, 10, device='cuda')
out = torch.empty(B * Q, 256, device='cuda')
rng = (ctypes.c_double * 6)(*S.PC_RANGE)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fn = lambda: lib.sbev_sasa_f32(p(qkvt), 776, p(bbox), rng, None, p(out), B, Q, H, 32, st)
for _ in range(5):
    assert fn() == 0
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for s, e in evs:
    s.record(); fn(); e.record()
torch.cuda.synchronize()
ts = sorted(s.elapsed_time(e) for s, e in evs)
print('sasa B=%d Q=%d: median %.1f us  min %.1f us' % (B, Q, ts[25] * 1e3, ts[0] * 1e3))
