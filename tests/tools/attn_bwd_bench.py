import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mergerec_amd import ops
DEV="cuda:0"
H=12
def run(lens, tag, sort=False):
    B=len(lens); T=sum(lens)
    cu=torch.tensor([0]+list(torch.tensor(lens).cumsum(0)),dtype=torch.int32,device=DEV)
    g=torch.Generator(device=DEV).manual_seed(0)
    qkv=torch.randn(T,3*H*64,device=DEV,generator=g)*0.5
    dctx=torch.randn(T,H*64,device=DEV,generator=g)
    ctx=ops.attention(qkv,cu,B,H,max(lens),products=0)
    order=torch.argsort(torch.tensor(lens),descending=True,stable=True).to(torch.int32).to(DEV) if sort else None
    for _ in range(3): ops.attention_bwd(qkv,ctx,dctx,cu,B,H,max_len=max(lens),seq_order=order)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): ops.attention_bwd(qkv,ctx,dctx,cu,B,H,max_len=max(lens),seq_order=order)
        torch.cuda.synchronize()
    pairs=sum(((l+31)//32)**2 for l in lens)*H
    for e in prof.key_averages():
        if "attn_bwd" in e.key:
            ms=e.device_time_total/5e3
            mf = 128 if "kv" in e.key else (32 if "<true" in e.key else 96)
            ideal=pairs*mf*64/1024/2.3e9*1e3
            print(f"{tag:10s} {e.key[:60]:60s} {ms:.3f} ms  ideal {ideal:.3f}  util {ideal/ms:.2f}")
run([512]*96,"uniform512")
run([128]*384,"uniform128")
import random
random.seed(1)
from mergerec_amd.synthetic import blair_item_lengths, blair_sequence_lengths
gg=torch.Generator().manual_seed(1)
lens=[int(x) for x in torch.cat([blair_sequence_lengths(64,gg),blair_item_lengths(64,gg)])]
run(lens,"ragged")
run(lens,"ragged-LPT",sort=True)
