import os, sys
sys.path.insert(0, os.getcwd())
import torch
from eetq_amd import ops
dev="cuda:0"
M,K,N=128,4096,4096
g=torch.Generator(device=dev); g.manual_seed(5)
w=((torch.rand(K,N,device=dev,generator=g)*2-1)/K**0.5).half()
raw,qw,sc=ops.quant_weights(w,torch.int8,True)
wdq=(raw.float()*sc.float()[None,:]).half().float()
x=(torch.rand(M,K,device=dev,generator=g)-0.25).half()
y=ops.w8_a16_gemm(x,qw,sc,path="tilesplit").float()
ref=x.float()@wdq
S=4; KT=K//64
parts=[x[:, (KT*s//S)*64:(KT*(s+1)//S)*64].float()@wdq[(KT*s//S)*64:(KT*(s+1)//S)*64] for s in range(S)]
print("err vs full", float((y-ref).abs().max()))
for s in range(S): print("err vs slice",s, float((y-parts[s]).abs().max()))
import itertools
for r in range(1,S+1):
    for c in itertools.combinations(range(S),r):
        e=float((y-sum(parts[i] for i in c)).abs().max())
        if e<0.05: print("matches sum of slices",c,e)
bad=((y-ref).abs()>0.05)
print("bad frac",float(bad.float().mean()),"rows bad:",bad.any(1).sum().item(),"cols bad:",bad.any(0).sum().item())
rb=bad.any(1).nonzero().flatten().tolist(); cb=bad.any(0).nonzero().flatten().tolist()
print(rb[:20], cb[:40])
# per tile fraction
bt=bad.view(M,N//64,64).any(2).any(0)
print("bad tiles", bt.sum().item(), "of", N//64)
