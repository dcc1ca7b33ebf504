import sys, torch
sys.path.insert(0, '/root/repo')
from cirkit_amd.circuit import HipCircuit
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data
from conftest import load_case
def labels(plan, tensors, B, x):
    hc = HipCircuit(plan, tensors, device='cuda:0')
    y = hc(x)
    rows = hc.profile_kernels(x, 3)
    ks = {}
    for r in rows: ks[r['kernel'].split('(')[0]] = ks.get(r['kernel'].split('(')[0], 0) + 1
    return ks
plan, tensors, g = load_case("cfg1_rbt8")
import numpy as np
x = torch.from_numpy(g["x"].astype(np.int64)).cuda()
print("cfg1 B=%d" % x.shape[0], labels(plan, tensors, x.shape[0], x))
plan = image_data((1,28,28), "quad-tree-2", num_input_units=32, num_sum_units=32)
t = init_plan_tensors(plan)
for B in (64, 256, 1024):
    print("cfg2 B=%d" % B, labels(plan, t, B, torch.randint(0,256,(B,784)).cuda()))
