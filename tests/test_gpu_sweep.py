"""A seeded sweep over the plan builders: image shapes, region graphs, sum-product layers, input layers, unit
counts (multiples of 32 and not), class counts and batch sizes (ragged tiles) -- every combination is evaluated by
the HIP path with its default settings (fusion, padding, hipGraph) and compared with the CPU oracle on the same
plan, parameters and inputs."""
import itertools
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
REL = 1e-4


def _cases():
    rng = np.random.default_rng(20250928)
    shapes = [(1, 4, 4), (1, 6, 6), (1, 8, 8), (3, 4, 4), (1, 5, 7), (1, 12, 12)]
    rgs = ["quad-tree-2", "quad-tree-4", "quad-graph", "random-binary-tree", "poon-domingos"]
    sps = ["cp", "cp-t", "tucker"]
    ins = ["categorical", "gaussian"]
    ks = [2, 7, 32, 33, 64]
    out = []
    for i in range(40):
        shape = shapes[rng.integers(len(shapes))]
        rg, sp, inp = rgs[rng.integers(len(rgs))], sps[rng.integers(len(sps))], ins[rng.integers(len(ins))]
        k = ks[rng.integers(len(ks))]
        if sp == "tucker" and (k > 32 or rg == "quad-tree-4"):  # K^arity weights: keep the oracle fast
            k = min(k, 7)
        ncls = int(rng.choice([1, 1, 1, 3, 10]))
        b = int(rng.choice([1, 31, 32, 100, 257]))
        mix = bool(rng.integers(2))
        out.append((i, shape, rg, sp, inp, k, ncls, b, mix))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"{c[0]}-{c[2]}-{c[3]}-{c[4]}-k{c[5]}-c{c[6]}-b{c[7]}")
def test_sweep_matches_oracle(hip_device, case):
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data
    from oracle.torch_oracle import as_torch, evaluate_plan

    i, shape, rg, sp, inp, k, ncls, b, mix = case
    try:
        plan = image_data(shape, rg, input_layer=inp, num_input_units=k, sum_product_layer=sp, num_sum_units=k,
                          num_classes=ncls, use_mixing_weights=mix)
    except (ValueError, NotImplementedError) as e:  # combinations the reference rejects as well
        pytest.skip(str(e))
    tensors = init_plan_tensors(plan, seed=i)
    g = torch.Generator().manual_seed(i)
    d = plan.num_variables
    x = torch.randn((b, d), generator=g) if inp == "gaussian" else torch.randint(0, 256, (b, d), generator=g)
    want = evaluate_plan(plan, as_torch(tensors), x)
    hc = HipCircuit(plan, tensors, device=hip_device)
    got = hc(x.to(hip_device)).cpu()
    assert got.shape == want.shape
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= REL * max(1.0, float(want.abs().max()))
    got2 = hc(x.to(hip_device)).cpu()  # replay of the recorded graph
    assert torch.equal(got, got2)
    # a marginal query over a random subset of the variables (IntegrateQuery, queries.py:53-131)
    mask = torch.rand(d, generator=g) < 0.4
    want_m = evaluate_plan(plan, as_torch(tensors), x, integrate_mask=mask.unsqueeze(0).expand(b, d))
    got_m = hc(x.to(hip_device), integrate_vars=mask).cpu()
    assert float((got_m - want_m).abs().max()) <= REL * max(1.0, float(want_m.abs().max()))
