"""`cirkit_amd.padding.pad_units`: the padded plan computes the same function (checked with the CPU oracle),
parameter values convert both ways, and unsupported plans are left alone."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.padding import pad_tensors, pad_units  # noqa: E402
from cirkit_amd.templates import image_data, tabular_data  # noqa: E402
from oracle.torch_oracle import as_torch, evaluate_plan  # noqa: E402

CASES = [
    dict(region_graph="quad-tree-2", sum_product_layer="cp"),
    dict(region_graph="quad-tree-4", sum_product_layer="cp-t"),
    dict(region_graph="quad-graph", sum_product_layer="cp"),
    dict(region_graph="quad-graph", sum_product_layer="tucker"),
    dict(region_graph="poon-domingos", sum_product_layer="cp", input_layer="gaussian"),
    dict(region_graph="random-binary-tree", sum_product_layer="cp", use_mixing_weights=False),
]


@pytest.mark.parametrize("kw", CASES, ids=[f"{c['region_graph']}-{c['sum_product_layer']}" for c in CASES])
@pytest.mark.parametrize("K", [6, 40])
def test_padded_plan_is_the_same_function(kw, K):
    if K == 40 and kw["sum_product_layer"] == "tucker":
        pytest.skip("oracle too slow")
    plan = image_data((1, 4, 4), num_input_units=K, num_sum_units=K, **kw)
    res = pad_units(plan)
    assert res is not None
    padded, info = res
    assert all(l.num_output_units in (1, 32, 64) and l.num_input_units in (1, 32, 64) for l in padded.layers)
    tensors = init_plan_tensors(plan)
    ptens = pad_tensors(info, tensors)
    for name, (shape, _) in padded.tensors.items():
        assert tuple(ptens[name].shape) == tuple(shape)
        assert np.array_equal(info.unpad(name, ptens[name]), tensors[name])
    g = torch.Generator().manual_seed(1)
    if kw.get("input_layer") == "gaussian":
        x = torch.randn((9, 16), generator=g)
    else:
        x = torch.randint(0, 256, (9, 16), generator=g)
    want = evaluate_plan(plan, as_torch(tensors), x)
    got = evaluate_plan(padded, as_torch(ptens), x)[..., : info.out_units]
    assert got.shape == want.shape
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_multiclass_output_is_sliced():
    plan = tabular_data(6, "random-binary-tree", input_layers=[{"name": "categorical", "args": {"num_categories": 3}}],
                        num_input_units=5, num_sum_units=5, num_classes=3) if False else None
    plan = image_data((1, 4, 4), "quad-tree-2", num_input_units=5, num_sum_units=5, num_classes=3)
    padded, info = pad_units(plan)
    assert info.out_units == 3
    tensors = init_plan_tensors(plan)
    x = torch.randint(0, 256, (4, 16))
    want = evaluate_plan(plan, as_torch(tensors), x)
    got = evaluate_plan(padded, as_torch(pad_tensors(info, tensors)), x)[..., :3]
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("name", ["quadgraph_6x6_k4", "pd_gauss_6x6_k4", "cfg1_rbt8", "plan_quadgraph_1x8x8_cp_mixFalse",
                                  "binomial_qg6x6_k4"])
def test_reference_plans_with_mixed_parameter_graphs(name):
    """Plans extracted from the reference (folds of one layer built from DIFFERENT parameter graphs: plain mixing
    weights next to collapsed Sum -> Sum products) pad to the same function."""
    from conftest import GOLDEN, load_case
    from cirkit_amd.plan import Plan

    if name.startswith("plan_"):  # plan-only fixture
        plan = Plan.load(os.path.join(GOLDEN, name))
        tensors = init_plan_tensors(plan, seed=1)
    else:
        plan, tensors, _ = load_case(name)
    res = pad_units(plan)
    assert res is not None
    padded, info = res
    gen = torch.Generator().manual_seed(2)
    gauss = any(l.type == "gaussian" for l in plan.layers)
    x = torch.randn((7, plan.num_variables), generator=gen) if gauss else torch.randint(0, 2, (7, plan.num_variables), generator=gen)
    want = evaluate_plan(plan, as_torch(tensors), x)
    got = evaluate_plan(padded, as_torch(pad_tensors(info, tensors)), x)[..., : info.out_units]
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_nothing_to_do_or_unsupported():
    assert pad_units(image_data((1, 4, 4), "quad-tree-2", num_input_units=32, num_sum_units=32)) is None
    sq = image_data((1, 4, 4), "quad-tree-2", input_layer="embedding", num_input_units=6, sum_product_layer="cp-t",
                    num_sum_units=6, sum_weight_activation="none", semiring="complex-lse-sum")
    assert pad_units(sq) is None


def test_unconstrained_parameters():
    """Raw (activation-free) parameters: padded weights are 0, a padded Gaussian keeps a positive stddev."""
    from conftest import load_case

    for name in ("kat_gaussian_f1o1", "kat_bernoulli_f1o1"):
        plan, tensors, g = load_case(name)
        padded, info = pad_units(plan)
        x = torch.from_numpy(np.asarray(g["x"]))
        want = evaluate_plan(plan, as_torch(tensors), x)
        got = evaluate_plan(padded, as_torch(pad_tensors(info, tensors)), x)[..., : info.out_units]
        assert torch.isfinite(got).all()
        assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


def test_padded_input_units_never_dominate():
    """Inputs far in the tails: every real unit of a Gaussian / Binomial layer is astronomically unlikely.  Padded
    input units are copies of real ones, so the padded circuit still returns the finite value of the original
    (a dummy unit, e.g. a standard Gaussian, would win every row maximum and underflow the real units to log 0)."""
    for kw, x in ((dict(input_layer="gaussian"), torch.full((3, 16), 40.0)),
                  (dict(input_layer="binomial"), torch.tensor([[0] * 16, [255] * 16, [128] * 16])),
                  (dict(input_layer="categorical"), torch.zeros((2, 16), dtype=torch.int64))):
        plan = image_data((1, 4, 4), "quad-tree-2", num_input_units=5, num_sum_units=5, **kw)
        tensors = init_plan_tensors(plan, seed=2)
        if kw["input_layer"] == "categorical":  # peaked distributions that give category 0 a probability ~ e^-95 (a denormal)
            name = plan.layers[0].params["probs"].nodes[0].config["tensor"]
            tensors[name] = tensors[name].copy()
            tensors[name][..., 0] -= 95.0
        padded, info = pad_units(plan)
        want = evaluate_plan(plan, as_torch(tensors), x)
        got = evaluate_plan(padded, as_torch(pad_tensors(info, tensors)), x)[..., : info.out_units]
        assert torch.isfinite(want).all() and torch.isfinite(got).all(), kw
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()), kw


def test_padded_sum_units_never_dominate():
    """Every real output of the first dense layer puts (almost) all its weight on an input unit that is e^-80
    unlikely at the observed values, while another input unit is likely.  A padded output unit that mixed the inputs
    uniformly would be ~80 nats above every real output and underflow them in the next layer; padded rows are
    copies of real rows, so the padded circuit returns the original's finite value."""
    plan = image_data((1, 4, 4), "quad-tree-2", num_input_units=2, num_sum_units=2)
    tensors = {k: v.copy() for k, v in init_plan_tensors(plan, seed=4).items()}
    cat = plan.layers[0].params["probs"].nodes[0].config["tensor"]
    dense = plan.layers[1].params["weight"].nodes[0].config["tensor"]
    tensors[cat][:, 0, :] = 0.0
    tensors[cat][:, 0, 7] = -80.0   # unit 0: category 7 has probability ~ e^-80 / 255
    tensors[cat][:, 1, :] = 0.0     # unit 1: uniform
    tensors[dense][:, :, 0] = 200.0  # every dense output: weight ~ 1 on unit 0, ~ e^-200 on unit 1
    tensors[dense][:, :, 1] = 0.0
    x = torch.full((2, 16), 7, dtype=torch.int64)
    padded, info = pad_units(plan)
    want = evaluate_plan(plan, as_torch(tensors), x)
    got = evaluate_plan(padded, as_torch(pad_tensors(info, tensors)), x)[..., : info.out_units]
    assert torch.isfinite(want).all() and torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
