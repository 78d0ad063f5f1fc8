import sys, os, torch
sys.path.insert(0, os.getcwd())
from mergerec_amd import ops
dev="cuda:0"; N=8; P=124645632//64*64
base=torch.randn(P,device=dev); tv=torch.randn(N,P,device=dev)*1e-3; alpha=torch.full((N,),0.125,device=dev); out=torch.empty(P,device=dev)
for _ in range(3): ops.merge_nway(base,tv,alpha,out=out)
torch.cuda.synchronize()
ts=[]
for _ in range(20):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record(); ops.merge_nway(base,tv,alpha,out=out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms=sorted(ts)[len(ts)//2]
print(f"merge_nway N=8 P={P/1e6:.1f}M: {ms:.4f} ms  {(N+2)*P*4/ms/1e9:.2f} TB/s")
