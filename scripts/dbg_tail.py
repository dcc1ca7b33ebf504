import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from cirkit_amd.circuit import HipCircuit
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data
mode = sys.argv[1]; B = int(sys.argv[2])
plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32, sum_product_layer="cp", num_sum_units=32)
t = init_plan_tensors(plan)
kw = dict(device="cuda:0", persistent_leaf=True)
x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(1)).to("cuda:0")
ref = HipCircuit(plan, t, merge_tail=False, **kw)
yr = ref(x).clone(); sr = ref.log_likelihood_sum(x).clone()
torch.cuda.synchronize(); print("ref ok", float(yr.mean()), flush=True)
hc = HipCircuit(plan, t, merge_tail=True, direct_input=(mode != "staged"), keep_layer_outputs=(mode != "nokeep"), **kw)
print("tail_in_leaf", hc._bind(B).tail_in_leaf, flush=True)
y = hc(x).clone(); torch.cuda.synchronize(); print("fwd ok", torch.equal(y, yr), float((y - yr).abs().max()), flush=True)
s = hc.log_likelihood_sum(x).clone(); torch.cuda.synchronize(); print("ll ok", torch.equal(s, sr), s.tolist(), sr.tolist(), flush=True)
