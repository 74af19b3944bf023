import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebev_amd import dense, synthetic as S
DEV='cuda:0'
def run(Q, use_mask, seed):
    g = torch.Generator().manual_seed(seed)
    B, H, D = 2, 8, 256
    x = torch.randn(B, Q, D, generator=g); bbox = torch.rand(B, Q, 10, generator=g)
    in_w, in_b = torch.randn(3*D, D, generator=g)/16, 0.1*torch.randn(3*D, generator=g)
    out_w, out_b = torch.randn(D, D, generator=g)/16, 0.1*torch.randn(D, generator=g)
    tau_w, tau_b = 0.02*torch.randn(H, D, generator=g), 2*torch.rand(H, generator=g)
    mask=None
    if use_mask:
        mask = torch.rand(Q, Q, generator=g) < 0.3; mask.fill_diagonal_(False)
    d = lambda t: t.to(DEV) if t is not None else None
    errs=[]
    xd = x.double(); cx = bbox[...,0].double()*102.4-51.2; cy = bbox[...,1].double()*102.4-51.2
    xy = torch.stack([cx,cy],-1); dist = -(xy[:,:,None]-xy[:,None]).norm(dim=-1)
    tau = xd@tau_w.double().t()+tau_b.double(); bias = dist[:,None]*tau.permute(0,2,1)[...,None]
    if mask is not None: bias = bias.masked_fill(mask[None,None], float('-inf'))
    qkv = xd@in_w.double().t()+in_b.double()
    q,k,v = (t.reshape(B,Q,H,32).permute(0,2,1,3) for t in qkv.chunk(3,-1))
    att = torch.softmax(q@k.transpose(-1,-2)/math.sqrt(32)+bias,-1)@v
    ref = xd + att.permute(0,2,1,3).reshape(B,Q,D)@out_w.double().t()+out_b.double()
    for it in range(5):
        y = dense.scale_adaptive_self_attention(d(bbox), d(x), S.PC_RANGE, H, d(in_w), d(in_b), d(out_w), d(out_b), d(tau_w), d(tau_b), d(mask))
        e=(y.cpu().double()-ref).abs()
        errs.append((e.max().item(), int((e>2e-5).sum())))
    print(Q, use_mask, seed, errs)
for Q,m,s in [(37,True,37),(37,True,1),(37,False,37),(100,True,100),(900,False,900),(64,True,5),(33,True,6)]:
    run(Q,m,s)
