import os,sys,time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torch
from cirkit_amd import _capi as capi
dev=torch.device("cuda:0")
B,K,C=4096,32,256
for F in (256,512,784,1024,1536):
    g=torch.randn(F//2 if F%2==0 else F, B, K, device=dev)   # pairs share a block like the trainer
    gfold=(torch.arange(F,device=dev,dtype=torch.int32)//2).contiguous()
    x=torch.randint(0,C,(F,B),device=dev,dtype=torch.int32)
    scope=torch.arange(F,dtype=torch.int64,device=dev)
    dt=torch.zeros(F,C+1,K,device=dev)
    s=torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        capi.call("ck_categorical_bwd", g.data_ptr(), gfold.data_ptr(), x.data_ptr(), scope.data_ptr(), dt.data_ptr(), F,B,K,C,0,None,s)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        capi.call("ck_categorical_bwd", g.data_ptr(), gfold.data_ptr(), x.data_ptr(), scope.data_ptr(), dt.data_ptr(), F,B,K,C,0,None,s)
    e1.record(); torch.cuda.synchronize()
    print(f"F={F}: {e0.elapsed_time(e1)/20*1e3:.1f} us  ({e0.elapsed_time(e1)/20*1e3/F*512:.1f} us per 512 folds)")
