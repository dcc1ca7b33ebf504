import os,sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cirkit_amd.plan import Plan
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.training_squared import HipSquaredTrainer
G=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"tests","golden")
plan_c=Plan.load(os.path.join(G,"cfg5_sos_c_k32"))
t=init_plan_tensors(plan_c); t={k:np.where(v==0,np.float32(1e-2),v).astype(np.float32) for k,v in t.items()}
B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
CH=int(sys.argv[2]) if len(sys.argv)>2 else 32
x=torch.randint(0,256,(B,784),generator=torch.Generator().manual_seed(1)).cuda()
res={}
for signed in (False, True):
    tr=HipSquaredTrainer(plan_c,t,device="cuda:0",lr=1e-3,signed=signed)
    tr.loss_and_grads(x); torch.cuda.synchronize()
    full=tr._flat_grad.double().clone()
    acc=torch.zeros_like(full)
    for i in range(0,B,CH):
        tr.loss_and_grads(x[i:i+CH].contiguous(), global_batch=B); torch.cuda.synchronize()
        acc+=tr._flat_grad.double()
    print("signed" if signed else "complex", "full vs sum of chunks: max abs diff", float((full-acc).abs().max()), "max", float(full.abs().max()), float(acc.abs().max()))
    res[signed]=(full,acc)
print("chunks complex vs signed", float((res[False][1]-res[True][1]).abs().max()))
print("full complex vs signed", float((res[False][0]-res[True][0]).abs().max()))
