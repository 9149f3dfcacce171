import torch, time
dev="cuda:0"
T,H,D=1024,40,128
qkv=torch.randn(1,T,3*H*D,dtype=torch.float16,device=dev)
q=qkv[...,:H*D].unflatten(-1,(H,D)); k=qkv[...,H*D:2*H*D].unflatten(-1,(H,D))
def tm(f,n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)*1e3/n
print("q.transpose(1,2).contiguous()  us", tm(lambda: q.transpose(1,2).contiguous()))
qc=q.contiguous()
print("contig [T,H,D] -> transpose.contiguous us", tm(lambda: qc.transpose(1,2).contiguous()))
out=torch.empty(1,H,T,D,dtype=torch.float16,device=dev)
print("copy_ into preallocated us", tm(lambda: out.copy_(q.transpose(1,2))))
print("permute via reshape/movedim us", tm(lambda: q.movedim(2,1).clone(memory_format=torch.contiguous_format)))
cache=torch.zeros(1,H,1152,D,dtype=torch.float16,device=dev)
idx=torch.arange(T,device=dev)
kt=k.transpose(1,2)
print("cache.index_copy_(2, idx, k^T) us", tm(lambda: cache.index_copy_(2, idx, kt)))
print("cache[:,:,:T]=k^T us", tm(lambda: cache[:,:,:T].copy_(kt)))
mask=torch.zeros(1,1,T,1152,dtype=torch.float16,device=dev)
qh=q.transpose(1,2).contiguous()
print("sdpa full cache + mask us", tm(lambda: torch.nn.functional.scaled_dot_product_attention(qh,cache,cache,attn_mask=mask)))
kk=cache[:,:,:T]
print("sdpa causal on T rows us", tm(lambda: torch.nn.functional.scaled_dot_product_attention(qh,kk,kk,is_causal=True)))
print("sdpa causal, non-contig q/k views us", tm(lambda: torch.nn.functional.scaled_dot_product_attention(q.transpose(1,2),kt,kt,is_causal=True)))
