"""Dev tool: what can fusing the layer tail buy?  (a) floor of a chain of DEPENDENT minimal launches on one
stream; (b) the real tail chain of a decoder layer at config 2 (FFN0, FFN1, norm3 | cls0 | reg0, LN | reg2, cls3 | reg4,
LN | refine, cls6 | next pos-encoder stage) timed as a chain, with the sum of its stand-alone kernel times beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sparsebev_amd import dense, synthetic as S
dev = 'cuda:0'
torch.set_grad_enabled(False)
M, D = 900, 256
x = torch.randn(M, D, device=dev)
g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
tiny = torch.randn(4, D, device=dev)

def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us per call

# (a) dependent minimal launches: y = LN(y) on 4 rows, chained
def chain_min(k=7):
    y = tiny
    for _ in range(k):
        y = dense.layer_norm(y, g, b)
    return y
t7 = timeit(chain_min)
print('chain of 7 dependent minimal launches: %.1f us  (%.2f us per launch)' % (t7, t7 / 7))

# (b) the tail's small linears at M = 900
w256 = torch.randn(D, D, device=dev) * 0.05
w512 = torch.randn(512, D, device=dev) * 0.05
w512b = torch.randn(D, 512, device=dev) * 0.05
w10 = torch.randn(10, D, device=dev) * 0.05
bb = torch.zeros(512, device=dev)
def tail():
    h = dense.linear(x, w512, bb, relu=True)
    t = dense.linear(h, w512b, bb[:D], residual=x)
    x3, c0 = dense.ln_linear(t, g, b, w256, bb[:D])
    r0 = dense.linear(x3, w256, bb[:D], relu=True)
    c1 = dense.layer_norm(c0, g, b, relu=True)
    r1 = dense.linear(r0, w256, bb[:D], relu=True)
    c2 = dense.linear(c1, w256, bb[:D])
    reg = dense.linear(r1, w10, bb[:10])
    c3 = dense.layer_norm(c2, g, b, relu=True)
    return dense.linear(c3, w10, bb[:10]), reg
print('tail as 10 stand-alone launches (op-by-op python path): %.1f us per chain' % timeit(tail, 100))
for name, fn in (('ffn0 [900,256]->512', lambda: dense.linear(x, w512, bb, relu=True)),
                 ('ffn1 [900,512]->256', lambda: dense.linear(torch.empty(M, 512, device=dev), w512b, bb[:D])),
                 ('linear 256->256', lambda: dense.linear(x, w256, bb[:D])),
                 ('ln_linear 256->256', lambda: dense.ln_linear(x, g, b, w256, bb[:D])),
                 ('layer_norm', lambda: dense.layer_norm(x, g, b)),
                 ('linear 256->10', lambda: dense.linear(x, w10, bb[:10]))):
    print('  %-22s %.1f us back-to-back (independent launches: throughput, not latency)' % (name, timeit(fn)))
