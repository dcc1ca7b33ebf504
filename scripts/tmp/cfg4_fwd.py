import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cirkit_amd.circuit import HipCircuit
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data
plan = image_data((1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64, sum_product_layer="cp", num_sum_units=64)
t = init_plan_tensors(plan)
x = torch.randn(4096, 784).cuda()
for ov in (False, True):
    hc = HipCircuit(plan, t, device="cuda:0", params_overlap=ov)
    for _ in range(5): hc(x)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20): hc(x)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 20)
    print("params_overlap", ov, "bound:", hc._bind(4096).params_overlap, "fork_after", hc._bind(4096).fork_after, "config 4 forward", round(sorted(ts)[2], 4), "ms", hc.num_launches(4096), "ops")
