import os,sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cirkit_amd.plan import Plan
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.training_squared import HipSquaredTrainer
G=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"tests","golden")
plan_c=Plan.load(os.path.join(G,"cfg5_sos_c_k32"))
t=init_plan_tensors(plan_c); t={k:np.where(v==0,np.float32(1e-2),v).astype(np.float32) for k,v in t.items()}
B=int(sys.argv[1]) if len(sys.argv)>1 else 4096
x=torch.randint(0,256,(B,784)).cuda()
for signed in (False, True):
    tr=HipSquaredTrainer(plan_c,t,device="cuda:0",lr=1e-3,signed=signed)
    for it in range(5):
        ll=tr.loss_and_grads(x)
        torch.cuda.synchronize()
        g=tr._flat_grad
        print(signed, it, float(ll[0]/ll[1]), "grad nan", int(torch.isnan(g).sum()), "inf", int(torch.isinf(g).sum()), float(g.abs().max()))
    if signed:
        sc=tr._signed; st=sc.bind(B)
        for i,k in sc.kind.items():
            if k=="emb": continue
            l=tr.c.layers[i]; o=st["off"][i]; n=l.num_folds*B*(32 if l.num_output_units==32 else l.num_output_units)
            a=st["arena"][o:o+n]; ga=st["garena"][o:o+l.num_folds*B*32]
            print(i,k,"val nan",int(torch.isnan(a).sum()),"inf",int(torch.isinf(a).sum()),"min",float(a.min()),"max",float(a.max()),"| g nan",int(torch.isnan(ga).sum()),"inf",int(torch.isinf(ga).sum()), float(ga.abs().max()))
    gs = {k: v.copy() for k,v in tr.gradients().items()}
    if not signed: ref=gs
    else:
        for k in gs: print(k, float(np.abs(gs[k]-ref[k]).max()), float(np.abs(ref[k]).max()))
