import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebev_amd import dense
for (N, K) in [(32768, 256), (256, 32768)]:
    x = torch.randn(900, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
    for _ in range(6):
        dense.linear(x, w, b)
torch.cuda.synchronize()
