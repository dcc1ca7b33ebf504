"""Which launch (kernel label) every layer of the committed fixtures / BASELINE shapes takes on the GPU box: python scripts/probe_labels.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cirkit_amd.circuit import HipCircuit
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data
from conftest import load_case
def labels(plan, tensors, x):
    hc = HipCircuit(plan, tensors, device='cuda:0')
    hc(x)
    ks = {}
    for r in hc.profile_kernels(x, 3):
        k = r['kernel'].split('(')[0]
        ks[k] = ks.get(k, 0) + 1
    return ks
for name in ("cfg1_rbt8", "cfg2_qt784", "cfg2t_qt784_cpt16", "cfg5_sos_c_k32"):
    plan, tensors, g = load_case(name)
    x = torch.from_numpy(g["x"].astype(np.float32 if g["x"].dtype.kind == "f" else np.int64)).cuda()
    print(name, "B=%d" % x.shape[0], labels(plan, tensors, x))
plan = image_data((1,28,28), "quad-tree-2", num_input_units=32, num_sum_units=32)
t = init_plan_tensors(plan)
for B in (16, 64, 256, 1024):
    print("cfg2 B=%d" % B, labels(plan, t, torch.randint(0,256,(B,784)).cuda()))
plan = image_data((1,8,8), "quad-tree-2", num_input_units=32, num_sum_units=32)
t = init_plan_tensors(plan)
for B in (64, 4096):
    print("qt 8x8 K=32 B=%d" % B, labels(plan, t, torch.randint(0,256,(B,64)).cuda()))
