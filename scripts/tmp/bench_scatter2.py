import os,sys
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
from cirkit_amd import _capi as capi
if os.environ.get('CK_LIB'): capi._LIB_PATH=os.path.join(ROOT,os.environ['CK_LIB'])
from cirkit_amd.plan import Plan
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.training_squared import HipSquaredTrainer
G=os.path.join(ROOT,"tests","golden")
plan_c=Plan.load(os.path.join(G,"cfg5_sos_c_k32"))
t=init_plan_tensors(plan_c); t={k:np.where(v==0,np.float32(1e-2),v).astype(np.float32) for k,v in t.items()}
B=4096
tr=HipSquaredTrainer(plan_c,t,device="cuda:0",lr=1e-3)
x=torch.randint(0,256,(B,784),generator=torch.Generator().manual_seed(B)).cuda()
for _ in range(4): tr.step(x)
torch.cuda.synchronize()
sc=tr._signed; st=sc.bind(B); c=tr.c
s=torch.cuda.current_stream().cuda_stream
def tm(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
ga=st["garena"].data_ptr()
for i,k in sc.kind.items():
    l=c.layers[i]
    if k=="emb":
        gfold,g,order=st["gfold"][i]
        print("gfold head", gfold[:8].tolist(), "order head", order[:20].tolist())
        f=lambda: capi.call("ck_embedding_bwd", ga+4*st["off"][g], 1, gfold.data_ptr(), order.data_ptr(), st["xt"].data_ptr(), l._scope(c.device).data_ptr(), l._table.data_ptr(), sc.grads[sc.wname[i]].data_ptr(), l.num_folds, B, 32, l.num_states, s)
        print("embedding_bwd in trainer buffers", tm(f))
        f2=lambda: capi.call("ck_embedding_bwd", ga+4*st["off"][g], 1, gfold.data_ptr(), None, st["xt"].data_ptr(), l._scope(c.device).data_ptr(), l._table.data_ptr(), sc.grads[sc.wname[i]].data_ptr(), l.num_folds, B, 32, l.num_states, s)
        print("  without fold order", tm(f2))
        print("scope head", l._scope(c.device)[:8].tolist())
# whole c list pieces
print("forward list", tm(lambda: sc.forward(B,s)))
print("backward list", tm(lambda: sc.backward(B,-2.0/B,s)))
