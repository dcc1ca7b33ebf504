import os,sys
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
from cirkit_amd import _capi as capi
F,B,K,C=784,4096,32,256
dev="cuda:0"
g=torch.Generator().manual_seed(0)
x=torch.randint(0,C,(B,F),generator=g)
xt=x.t().contiguous().to(torch.int32).to(dev)
scope=torch.arange(F,dtype=torch.int64,device=dev)
gout=torch.randn(F//2,B,K,generator=g).to(dev)
gfold=(torch.arange(F,dtype=torch.int32)//2).to(dev)
order=[]
for i0 in range(0,F//2,8):
    for m in range(2):
        order+=[2*(i0+j)+m for j in range(8)]
order=torch.tensor(order,dtype=torch.int32,device=dev)
table=(torch.rand(F,C+1,K,generator=g)+0.5).to(dev)
dt=torch.zeros(F,C+1,K,device=dev); dw=torch.zeros(F,K,C,device=dev)
s=torch.cuda.current_stream().cuda_stream
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
print("categorical_bwd gfold+order", t(lambda: capi.call("ck_categorical_bwd", gout.data_ptr(), gfold.data_ptr(), xt.data_ptr(), scope.data_ptr(), dt.data_ptr(), F,B,K,C,0,order.data_ptr(),s)))
print("categorical_bwd gfold      ", t(lambda: capi.call("ck_categorical_bwd", gout.data_ptr(), gfold.data_ptr(), xt.data_ptr(), scope.data_ptr(), dt.data_ptr(), F,B,K,C,0,None,s)))
print("embedding_bwd gfold+order  ", t(lambda: capi.call("ck_embedding_bwd", gout.data_ptr(), 1, gfold.data_ptr(), order.data_ptr(), xt.data_ptr(), scope.data_ptr(), table.data_ptr(), dw.data_ptr(), F,B,K,C,s)))
print("embedding_bwd gfold        ", t(lambda: capi.call("ck_embedding_bwd", gout.data_ptr(), 1, gfold.data_ptr(), None, xt.data_ptr(), scope.data_ptr(), table.data_ptr(), dw.data_ptr(), F,B,K,C,s)))
