import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from sparsebev_amd import _lib, dense
from test_gpu_bf16s import _qual_inputs
lib = _lib.load()
M, N, K = 900, 32768, 256
x, w, b, _ = _qual_inputs('outlier', M, N, K, 31)
ref = x.double() @ w.double().t() + b.double()
mag = x.double().abs() @ w.double().abs().t() + b.double().abs()
wf, wsc = dense.pack_f16s_frags(w)
for name, ws_on, nprod in (('ws f16x3', 1, 3), ('tiled f16x3', 0, 3), ('ws f16x4', 1, 4)):
    lib.sbev_linear_gen_weight_stationary(ws_on)
    y = dense.linear_f16s_gen(x, wf, wsc, b, nprod=nprod)
    r = ((y.double() - ref).abs() / mag)
    i = int(r.argmax()); rr, nn = i // N, i % N
    print(name, 'worst ratio %.3e at (%d, %d): y %.6e ref %.6e mag %.6e' % (r.max().item(), rr, nn, y[rr, nn].item(), ref[rr, nn].item(), mag[rr, nn].item()))
    xi, wi = int(x[rr].abs().argmax()), int(w[nn].abs().argmax())
    print('   x outlier idx %d val %.4e; w outlier idx %d val %.4e; x[wi] %.4e w[xi] %.4e; xmax tensor %.4e' % (xi, x[rr, xi].item(), wi, w[nn, wi].item(), x[rr, wi].item(), w[nn, xi].item(), x.abs().max().item()))
lib.sbev_linear_gen_weight_stationary(1)
yf = dense.linear(x, w, b)
print('exact worst ratio %.3e' % ((yf.double() - ref).abs() / mag).max().item())
# per-element representation check of the packed images
xs, xsc = dense.pack_f16s_frags(x, per_tensor=True)
print('x scale', xsc.tolist())
