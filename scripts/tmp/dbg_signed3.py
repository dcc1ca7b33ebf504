import os,sys
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
from cirkit_amd.plan import Plan
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.training_squared import HipSquaredTrainer
from oracle import torch_oracle as oracle
from cirkit_amd import _capi as capi
if os.environ.get('CK_LIB'): capi._LIB_PATH=os.path.join(ROOT,os.environ['CK_LIB'])
G=os.path.join(ROOT,"tests","golden")
plan_c=Plan.load(os.path.join(G,"cfg5_sos_c_k32")); plan_z=Plan.load(os.path.join(G,"cfg5_sos_z_k32"))
t=init_plan_tensors(plan_c); t={k:np.where(v==0,np.float32(1e-2),v).astype(np.float32) for k,v in t.items()}
B,CH=256,32
x=torch.randint(0,256,(B,784),generator=torch.Generator().manual_seed(1))
trs={s:HipSquaredTrainer(plan_c,t,plan_z=plan_z,device="cuda:0",signed=s) for s in (False,True)}
worst=(0,None)
for i in range(0,B,CH):
    g={}
    for s,tr in trs.items():
        tr.loss_and_grads(x[i:i+CH].cuda().contiguous()); torch.cuda.synchronize()
        g[s]=tr._flat_grad.double().clone()
    d=float((g[False]-g[True]).abs().max())
    print(i, d)
    if d>worst[0]: worst=(d,i)
i=worst[1]
xc=x[i:i+CH]
leaves={k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.float64).requires_grad_(True) for k,v in t.items()}
c=oracle.evaluate_plan(plan_c, leaves, xc, grad=True); z=oracle.evaluate_plan(plan_z, leaves, None, grad=True)
loss=-(2.0*c.real - z.real).mean(); loss.backward()
for s,tr in trs.items():
    tr.loss_and_grads(xc.cuda().contiguous()); torch.cuda.synchronize()
    got=tr.gradients()
    print("signed" if s else "complex", {k: (float(np.abs(got[k]-leaves[k].grad.numpy()).max()), float(np.abs(leaves[k].grad.numpy()).max())) for k in ("t0","t1","t5","t10")})
# which rows: per-row check of c(x) signs
print("c real min/max", float(c.real.min()), float(c.real.max()), "imag", c.imag.flatten()[:8])
