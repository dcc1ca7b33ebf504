import os,sys,time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cirkit_amd.plan import Plan
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.training_squared import HipSquaredTrainer
G=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"tests","golden")
plan_c=Plan.load(os.path.join(G,"cfg5_sos_c_k32"))
t=init_plan_tensors(plan_c); t={k:np.where(v==0,np.float32(1e-2),v).astype(np.float32) for k,v in t.items()}
tr=HipSquaredTrainer(plan_c,t,device="cuda:0",lr=1e-3,signed={"0":False,"1":True}.get(os.environ.get("CK_SIGNED"),None),use_graph=os.environ.get("CK_SQ_GRAPH","0")=="1")
for B in ([int(a) for a in sys.argv[1:]] or [256,4096]):
    x=torch.randint(0,256,(B,784),generator=torch.Generator().manual_seed(B)).cuda()
    for _ in range(3): ll=tr.step(x)
    torch.cuda.synchronize(); t0=time.perf_counter()
    n=int(os.environ.get("CK_SQ_STEPS","10"))
    host=[]
    for _ in range(n):
        h0=time.perf_counter(); ll=tr.step(x); host.append(time.perf_counter()-h0)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
    print("host time per step() call, us:", " ".join(f"{1e6*h:.0f}" for h in host[:12]))
    print(f"squared-circuit training step (config 5 plans, B={B}): {dt*1e3:.2f} ms, mean LL {float(ll[0]/ll[1]):.3f}")
