import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from eetq_amd import ops
from sweep import chain_us
dev="cuda:0"
for (M,K,N) in [(1024,5120,5120),(1024,13824,5120),(1024,4096,4096),(512,5120,5120),(2048,5120,5120)]:
    nbuf=max(2,(640<<20)//(K*N))
    g=torch.Generator(device=dev); g.manual_seed(1)
    sets=[tuple(ops.quant_weights(((torch.rand(K,N,device=dev,generator=g)*2-1)/K**0.5).half(),torch.int8,False)) for _ in range(nbuf)]
    x=torch.rand(M,K,device=dev,generator=g).half()
    t=chain_us(lambda i: ops.w8_a16_gemm(x,sets[i%nbuf][0],sets[i%nbuf][1]), nbuf)
    print(json.dumps({"M":M,"K":K,"N":N,"us":round(t,2),"TF":round(2.0*M*N*K/t/1e6,1)}),flush=True)
    del sets; torch.cuda.empty_cache()
