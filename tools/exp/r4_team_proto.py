"""Round-4 experiment: three chained row-local Linears (ffn.0 -> ffn.1 + residual -> norm3 -> [cls.0 | reg.0]) on teams of 8 workgroups
(csrc/row_team.hip) -- correctness against torch fp64 and the time per launch in the team-safe and same-XCD hand-off modes."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebev_amd import _lib, dense
_lib.load()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'librow_team.so'))
lib.sbev_last_error = ctypes.CDLL(_lib.LIB_PATH).sbev_last_error
lib.sbev_last_error.restype = ctypes.c_char_p
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 900
g = torch.Generator().manual_seed(5)
dev = 'cuda'
x2 = torch.randn(M, 256, generator=g).to(dev)
W0, b0 = (torch.randn(512, 256, generator=g) / 16).to(dev), torch.randn(512, generator=g).to(dev)
W1, b1 = (torch.randn(256, 512, generator=g) / 22).to(dev), torch.randn(256, generator=g).to(dev)
g3, be3 = (1 + 0.1 * torch.randn(256, generator=g)).to(dev), (0.1 * torch.randn(256, generator=g)).to(dev)
W2, b2 = (torch.randn(512, 256, generator=g) / 16).to(dev), torch.randn(512, generator=g).to(dev)
packs = [dense.pack_f16s_frags(w) for w in (W0, W1, W2)]
h = torch.empty(M, 512, device=dev); u = torch.empty(M, 256, device=dev); x3 = torch.empty(M, 256, device=dev); y = torch.empty(M, 512, device=dev)
teams = (M + 31) // 32
counters = torch.zeros(teams * 4, device=dev, dtype=torch.int32)
err = torch.zeros(1, device=dev, dtype=torch.int32)


trace = torch.zeros(64, device=dev, dtype=torch.int64)


def run(safe, tr=False):
    return lib.sbev_row_team_proto(p(x2), p(packs[0][0]), p(packs[0][1][1].contiguous()), p(b0), p(packs[1][0]), p(packs[1][1][1].contiguous()), p(b1),
                                   p(g3), p(be3), p(packs[2][0]), p(packs[2][1][1].contiguous()), p(b2), p(h), p(u), p(x3), p(y), p(counters), p(err),
                                   ctypes.c_int64(M), int(safe), p(trace) if tr else ctypes.c_void_p(0), st)


xd = x2.double()
hr = torch.relu(xd @ W0.double().t() + b0.double())
ur = xd + hr @ W1.double().t() + b1.double()
x3r = torch.nn.functional.layer_norm(ur, (256,), g3.double(), be3.double(), 1e-5)
yr = x3r @ W2.double().t() + b2.double()
yr[:, 256:] = torch.relu(yr[:, 256:])
for safe in (1, 0):
    for t in (h, u, x3, y):
        t.fill_(float('nan'))
    rc = run(safe)
    torch.cuda.synchronize()
    assert rc == 0, lib.sbev_last_error()
    print('mode %s: error word %d, counters nonzero %d;  max err h %.2e  u %.2e  x3 %.2e  y %.2e' % (
        'safe (sc0 sc1)' if safe else 'same-XCD fast', int(err.item()), int((counters != 0).sum()),
        (h.double() - hr).abs().max().item(), (u.double() - ur).abs().max().item(), (x3.double() - x3r).abs().max().item(), (y.double() - yr).abs().max().item()))
    # repeated launches: the counters reset themselves
    for _ in range(20):
        run(safe)
    torch.cuda.synchronize()
    print('   after 20 more launches: max err y %.2e, error word %d' % ((y.double() - yr).abs().max().item(), int(err.item())))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for s_, e_ in ev:
        s_.record(); run(safe); e_.record()
    torch.cuda.synchronize()
    ts = sorted(s_.elapsed_time(e_) for s_, e_ in ev)
    # back-to-back launches without event gaps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        run(safe)
    e1.record(); torch.cuda.synchronize()
    print('   time per launch: median %.1f us (HIP events around each), %.1f us (100 back to back)' % (ts[len(ts) // 2] * 1e3, e0.elapsed_time(e1) * 10))

for safe in (1, 0):
    trace.zero_()
    run(safe, True); torch.cuda.synchronize()
    t = trace.cpu().tolist()
    names = ['W issued', 'team wait', 'X loaded', 'converted', 'barrier', 'MFMA', 'reduce+stores', 'arrive/barrier']
    print('mode %s: start -> placement barrier %d cycles' % ('safe' if safe else 'fast', t[1] - t[0]))
    for sidx in range(3):
        b = 2 + 8 * sidx
        prev = t[b - 1]
        print('   stage %d: ' % sidx + '  '.join('%s %d' % (n, t[b + i] - (prev if i == 0 else t[b + i - 1])) for i, n in enumerate(names)) + '   total %d' % (t[b + 7] - prev))
