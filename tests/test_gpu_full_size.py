"""BASELINE.json's configurations at their FULL size (batch 4096, plans built natively, closed-form parameters) on the
HIP path with default settings, through properties that do not need a reference run at that size:
chunk consistency (the same rows in one 4096-row batch and in 64-row batches), determinism of the replayed graph,
normalisation (all variables marginalised -> log 1), the sum rule on one variable, plus the CPU oracle on a slice."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
B = 4096


def _circuit(cfg, device):
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    if cfg == 2:
        plan = image_data((1, 28, 28), "quad-tree-2", input_layer="categorical", num_input_units=32,
                          sum_product_layer="cp", num_sum_units=32)
    elif cfg == 4:
        plan = image_data((1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64,
                          sum_product_layer="cp", num_sum_units=64)
    else:
        plan = image_data((1, 28, 28), "quad-tree-2", input_layer="embedding", num_input_units=32, sum_product_layer="cp-t",
                          num_sum_units=32, sum_weight_activation="none", semiring="complex-lse-sum")
    tensors = init_plan_tensors(plan)
    g = torch.Generator().manual_seed(100 + cfg)
    x = torch.randn((B, 784), generator=g) if cfg == 4 else torch.randint(0, 256, (B, 784), generator=g)
    return plan, tensors, x, HipCircuit(plan, tensors, device=device)


@pytest.mark.parametrize("cfg", [2, 4, 5])
def test_full_batch_is_consistent_deterministic_and_matches_the_oracle(hip_device, cfg):
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, x, hc = _circuit(cfg, hip_device)
    xd = x.to(hip_device)
    y = hc(xd).clone()
    assert y.shape[0] == B and bool(torch.isfinite(torch.view_as_real(y) if y.is_complex() else y).all())
    assert torch.equal(hc(xd), y)  # graph replay: bit-identical
    # the same rows evaluated 64 at a time (other tile counts, other workgroup mappings)
    for lo in (0, 1984, 4032):
        part = hc(xd[lo : lo + 64])
        ref = y[lo : lo + 64]
        err = (part - ref).abs().max()
        assert float(err) <= 2e-4 * float(ref.abs().max()), (cfg, lo, float(err))
    # a slice against the CPU oracle
    rows = slice(1000, 1000 + (64 if cfg == 2 else 16))
    want = evaluate_plan(plan, as_torch(tensors), x[rows])
    got = y[rows].cpu()
    if cfg == 5:
        e5 = float((got.real - want.real).abs().max()) / float(want.real.abs().max())
        print(f"\n[config 5, full size] max rel err of Re c(x) against the oracle slice: {e5:.2e}")
        assert e5 <= 1e-4  # (north_star's bar, as for the real configurations)
    else:
        assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("cfg", [2, 4])
def test_full_batch_normalisation_and_sum_rule(hip_device, cfg):
    plan, tensors, x, hc = _circuit(cfg, hip_device)
    xd = x.to(hip_device)
    # every variable marginalised: a normalised circuit integrates to 1
    allz = hc(xd, integrate_vars=range(784))
    assert float(allz.abs().max()) <= 2e-3
    if cfg != 2:
        return
    # sum rule on one variable: sum_c p(x_{-v}, x_v = c) = p(x_{-v}), 16 rows x 256 categories = one 4096-row batch
    v = 391
    base = x[:16].clone()
    grid = base.repeat_interleave(256, dim=0)
    grid[:, v] = torch.arange(256).repeat(16)
    joint = hc(grid.to(hip_device)).reshape(16, 256).double()
    marg = hc(base.to(hip_device), integrate_vars=[v]).reshape(16).double()
    assert float((torch.logsumexp(joint, dim=1) - marg).abs().max()) <= 1e-4 * float(marg.abs().max())


def test_many_batch_sizes_and_long_replay(hip_device):
    """Serving pattern: batch sizes change from call to call (the circuit keeps a few bindings and re-binds the
    rest), results must not depend on it; and a long run of replays neither drifts nor grows memory."""
    plan, tensors, x, hc = _circuit(2, hip_device)
    xd = x.to(hip_device)
    full = hc(xd).clone()
    g = torch.Generator().manual_seed(0)
    for _ in range(40):
        b = int(torch.randint(1, 4097, (1,), generator=g))
        y = hc(xd[:b])
        assert y.shape[0] == b
        assert float((y - full[:b]).abs().max()) <= 2e-4 * float(full[:b].abs().max())
    ref = hc.log_likelihood_sum(xd).clone()
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    for _ in range(2000):
        r = hc.log_likelihood_sum(xd)
    torch.cuda.synchronize()
    assert torch.equal(r, ref)
    assert torch.cuda.memory_allocated() == m0


def test_config3_per_node_batch_on_one_gpu(hip_device):
    """BASELINE config 3's global batch (32768 rows = 8 shards of 4096) evaluated on ONE GPU: the eight 4096-row shards a
    data-parallel run would hold give the same rows as the single 32768-row batch, the [sum, count] pairs of the shards
    add up to the pair of the whole (what the one all-reduce of the N-GPU run computes), and a slice matches the oracle."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan = image_data((1, 28, 28), "quad-tree-2", input_layer="categorical", num_input_units=32,
                      sum_product_layer="cp", num_sum_units=32)
    tensors = init_plan_tensors(plan)
    G = 32768
    x = torch.randint(0, 256, (G, 784), generator=torch.Generator().manual_seed(3))
    xd = x.to(hip_device)
    hc = HipCircuit(plan, tensors, device=hip_device)
    y = hc(xd).clone()
    assert y.shape == (G, 1, 1) and bool(torch.isfinite(y).all())
    whole = hc.log_likelihood_sum(xd).clone().cpu()
    assert whole[1].item() == G
    assert abs(whole[0].item() - float(y.double().sum())) <= 1e-9 * abs(float(y.double().sum()))
    tot = torch.zeros(2, dtype=torch.float64)
    for r in range(8):  # rank r of 8 holds rows [4096 r, 4096 (r + 1))
        shard = xd[4096 * r : 4096 * (r + 1)]
        ys = hc(shard)
        assert float((ys - y[4096 * r : 4096 * (r + 1)]).abs().max()) <= 2e-4 * float(y.abs().max())
        tot += hc.log_likelihood_sum(shard).cpu()
    assert tot[1].item() == G
    assert abs(tot[0].item() - whole[0].item()) <= 1e-6 * abs(whole[0].item())
    rows = slice(20000, 20064)
    want = evaluate_plan(plan, as_torch(tensors), x[rows])
    assert float((y[rows].cpu() - want).abs().max()) <= 1e-4 * float(want.abs().max())
