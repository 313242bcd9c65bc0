import sys; sys.path.insert(0,'/root/repo')
import torch
from nwhead_amd import ops
from oracle import nw_oracle as O
dev=torch.device('cuda:0')
g=torch.Generator().manual_seed(0)
B,N,d,C=96,3000,512,200
q=(torch.randn(B,d,generator=g)*3+0.5).to(dev); s=torch.randn(N,d,generator=g).to(dev)
sy=(torch.arange(N)%C).sort().values.to(dev)
cache=ops.SplitBank(s)
for kind in ("dotproduct","euclidean"):
    fast=ops.nw_head(q,s,sy,C,kind,support_cache=cache).cpu().double()
    slow=ops.nw_head(q,s,sy,C,kind).cpu().double()
    ref=O.nw_head_f64(q.cpu(),s.cpu(),sy.cpu(),C,kind)
    t32=O.nw_head_f32(q.cpu(),s.cpu(),sy.cpu(),C,kind).double()
    print(kind,"fast-ref",(fast-ref).abs().max().item(),"slow-ref",(slow-ref).abs().max().item(),"torchf32-ref",(t32-ref).abs().max().item())
    sc_f=ops.nw_scores(q,s,kind).cpu().double(); sc_r=O.scores_f64(q.cpu(),s.cpu(),kind)
    print("   scores fp32-mfma err", (sc_f-sc_r).abs().max().item(), "score magnitude", sc_r.abs().max().item())
