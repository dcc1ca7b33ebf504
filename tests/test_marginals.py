"""Marginal queries (IntegrateQuery, cirkit/backend/torch/queries.py:19-184).

Pins: the reference's known answers for marginals and partition functions (SURVEY.md section 8 c:
mar (1,0,1,1,.) = 16.845, Z = 318.0; mar (0.3,.) = 23.528960785605985, Z = 44.0) and outputs of the
real reference's IntegrateQuery under random per-row masks (tests/golden/*_marg.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case


def _marg(name):
    with np.load(os.path.join(GOLDEN, name + "_marg.npz")) as z:
        return {k: z[k] for k in z.files}


def _x(m):
    x = m["x"]
    return torch.from_numpy(x.astype(np.float32 if x.dtype.kind == "f" else np.int64))


@pytest.mark.parametrize("name", ["kat_bernoulli_f1o1", "kat_gaussian_f1o1", "cfg1_rbt8", "cfg2_qt784"])
def test_oracle_marginals_match_reference_integrate_query(name):
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, _ = load_case(name)
    m = _marg(name)
    y = evaluate_plan(plan, as_torch(tensors), _x(m), integrate_mask=torch.from_numpy(m["mask"]))
    assert np.array_equal(y.numpy(), m["y"])


def test_oracle_reproduces_the_reference_marginal_known_answers():
    from oracle.torch_oracle import as_torch, evaluate_plan

    for name, mar, z in (("kat_bernoulli_f1o1", 16.845, 318.0), ("kat_gaussian_f1o1", 23.528960785605985, 44.0)):
        plan, tensors, _ = load_case(name)
        m = _marg(name)
        y = torch.exp(evaluate_plan(plan, as_torch(tensors), _x(m), integrate_mask=torch.from_numpy(m["mask"])).double()).reshape(-1)
        assert abs(float(y[0]) - mar) <= 1e-5 * mar  # row 0: last variable marginalised
        assert abs(float(y[1]) - z) <= 1e-5 * z  # row 1: everything marginalised = partition function
        assert float(m["kat_mar"][0]) == pytest.approx(mar) and float(m["kat_z"]) == z


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kat_bernoulli_f1o1", "kat_gaussian_f1o1", "cfg1_rbt8", "cfg2_qt784"])
@pytest.mark.parametrize("fuse", [False, True])
def test_hip_marginals_match_reference(hip_device, name, fuse):
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, _ = load_case(name)
    m = _marg(name)
    hc = HipCircuit(plan, tensors, device=hip_device, use_graph=False, fuse=fuse)
    mask = torch.from_numpy(m["mask"])
    y = hc(_x(m).to(hip_device), integrate_vars=mask.to(hip_device)).cpu()
    ref = torch.from_numpy(m["y"])
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    # an unmasked call afterwards is the plain likelihood again (the mask is not sticky)
    y0 = hc(_x(m).to(hip_device)).cpu()
    y1 = hc(_x(m).to(hip_device), integrate_vars=torch.zeros_like(mask).to(hip_device)).cpu()
    assert torch.equal(y0, y1)


@pytest.mark.gpu
def test_hip_marginal_known_answers_and_argument_forms(hip_device):
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, _ = load_case("kat_bernoulli_f1o1")
    hc = HipCircuit(plan, tensors, device=hip_device, use_graph=False)
    x = torch.tensor([[1, 0, 1, 1, 0]]).to(hip_device)
    assert abs(float(torch.exp(hc(x, integrate_vars=[4]))) - 16.845) <= 1e-3
    assert abs(float(torch.exp(hc(x, integrate_vars=range(5)))) - 318.0) <= 1e-2
    assert abs(float(torch.exp(hc(x, integrate_vars=torch.tensor([False, False, False, False, True])))) - 16.845) <= 1e-3
    with pytest.raises(ValueError):
        hc(x, integrate_vars=[7])
    with pytest.raises(ValueError):
        hc(x, integrate_vars=torch.ones(1, 5))
    with pytest.raises(ValueError):
        hc(x, integrate_vars=torch.ones(3, 5, dtype=torch.bool))
