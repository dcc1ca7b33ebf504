"""The CPU oracle against everything that pins it (no GPU):

* outputs of the REAL reference on the BASELINE configurations (tests/golden/*_golden.npz, written
  by tests/golden/make_fixtures.py which imports /root/reference) -- bit-for-bit in fp32 and fp64;
* the reference's own known-answer circuits (tests/symbolic/test_utils.py:293-503 of the reference:
  0.7626, 3.2266, Z = 318.0; 3.744904862456293, Z = 44.0) under all fold/optimize combinations;
* the reference's semiring underflow test (tests/backend/torch/test_semiring.py:41-61).
"""
import glob
import itertools
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case
from oracle.torch_oracle import _CLSE, _LSE, _einsum, as_torch, evaluate_plan


def _x_of(g):
    if "x" not in g:
        return None
    x = g["x"]
    return torch.from_numpy(x.astype(np.float32 if x.dtype.kind == "f" else np.int64))


@pytest.mark.parametrize("name", ["cfg1_rbt8", "cfg2_qt784", "cfg2t_qt784_cpt16", "cfg4_pd784", "tucker_qt16_k6",
                                  "tucker4_qt16_k3", "quadgraph_6x6_k4", "rbt6_perfeature_k2", "pd_gauss_6x6_k4"])
def test_oracle_is_bit_exact_vs_reference_fp32(name):
    plan, tensors, g = load_case(name)
    y = evaluate_plan(plan, as_torch(tensors), _x_of(g))
    assert y.shape == g["y_f32"].shape
    assert np.array_equal(y.numpy(), g["y_f32"])


@pytest.mark.parametrize("name", ["cfg1_rbt8", "cfg2_qt784"])
def test_oracle_is_bit_exact_vs_reference_fp64(name):
    plan, tensors, g = load_case(name)
    t64 = {k: v.double() for k, v in as_torch(tensors).items()}
    torch.set_default_dtype(torch.float64)
    try:
        y = evaluate_plan(plan, t64, _x_of(g))
    finally:
        torch.set_default_dtype(torch.float32)
    assert np.array_equal(y.numpy(), g["y_f64"])
    # the fp32 run agrees with fp64 to fp32 round-off (1.2e-7 at config 2)
    assert np.abs((g["y_f32"] - g["y_f64"]) / g["y_f64"]).max() < 1e-6


def test_oracle_complex_sos_circuit_and_partition_function():
    plan_c, tensors, gc = load_case("cfg5_sos_c_k32")
    plan_z, _, gz = load_case("cfg5_sos_z_k32")
    tt = as_torch(tensors)
    y = evaluate_plan(plan_c, tt, _x_of(gc))
    z = evaluate_plan(plan_z, tt, None)
    assert np.array_equal(y.numpy(), gc["y_c64"])
    assert np.array_equal(z.numpy(), gz["z_c64"])
    assert z.shape == (1, 1)  # empty scope: (O, K)
    # squared circuit: log p(x) = 2 Re c(x) - Re Z is a finite real number
    lp = 2 * y.real - z.real
    assert torch.isfinite(lp).all()


@pytest.mark.parametrize("name", ["sos_cat_c_qt4x4_k6", "sos_gauss_c_qt4x4_k4"])
def test_oracle_real_input_layers_under_the_complex_semiring(name):
    """Categorical / Gaussian log-likelihoods mapped into complex-lse-sum (layers/input.py:276-278)."""
    plan, tensors, g = load_case(name)
    y = evaluate_plan(plan, as_torch(tensors), _x_of(g))
    assert y.dtype == torch.complex64
    assert np.array_equal(y.numpy(), g["y_c64"])


KATS = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLDEN, "kat_*.json")))


@pytest.mark.parametrize("name", KATS)
def test_reference_known_answers(name):
    plan, tensors, g = load_case(name)
    x = _x_of(g)
    y = evaluate_plan(plan, as_torch(tensors), x)
    assert np.array_equal(y.numpy(), g["y_f32"])
    lin = torch.exp(y.double().reshape(-1))
    for kx, ky in zip(g["kat_x"], g["kat_y"]):
        row = [i for i in range(len(x)) if np.allclose(x[i].numpy(), kx)]
        assert row
        assert abs(float(lin[row[0]]) - float(ky)) <= 1e-4 * max(1.0, float(ky))
    if "bernoulli" in name:
        # exhaustive enumeration sums to the partition function (test_compile_circuit.py:27-51)
        assert len(x) == 32
        assert abs(float(lin.sum()) - float(g["kat_z"])) <= 1e-4 * float(g["kat_z"])


def test_fold_and_optimize_flags_do_not_change_the_function():
    """The reference's main internal invariant."""
    for fam in ("bernoulli", "gaussian"):
        ys = []
        for fo in ("f0o0", "f0o1", "f1o0", "f1o1"):
            plan, tensors, g = load_case(f"kat_{fam}_{fo}")
            ys.append(evaluate_plan(plan, as_torch(tensors), _x_of(g)).double())
        for y in ys[1:]:
            assert float((y - ys[0]).abs().max()) < 1e-5


def test_semiring_underflow_case():
    """tests/backend/torch/test_semiring.py:41-61 of the reference: lse and complex-lse einsum agree
    and stay finite where a naive exp would underflow in fp32."""
    a = torch.tensor([-200.0, -200.0, -5.0])
    b = torch.tensor([1.0, 2.0, 1e-38])
    r = _einsum(_LSE, "i,i->", inputs=(a,), operands=(b,), dim=0, keepdim=False)
    c = _einsum(_CLSE, "i,i->", inputs=(a.to(torch.complex64),), operands=(b,), dim=0, keepdim=False)
    assert torch.isfinite(r).all() and torch.isfinite(c.real).all()
    assert abs(float(r) - float(c.real)) < 1e-4
    # amax(-inf) is clamped to finfo.min: an all-zero row gives -inf, not NaN (semiring.py:392-399)
    z = torch.full((4,), float("-inf"))
    out = _einsum(_LSE, "i,i->", inputs=(z,), operands=(torch.ones(4),), dim=0, keepdim=False)
    assert float(out) == float("-inf")


def test_binomial_inputs_match_reference():
    """Binomial input layers (input.py:437-549): the oracle reproduces the reference outputs bit for bit, and the
    native `image_data(..., input_layer="binomial")` plan is the reference's."""
    import numpy as np
    import torch
    from conftest import load_case
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data
    from oracle.torch_oracle import as_torch, evaluate_plan
    from test_templates import _assert_same_plan

    plan, _, g = load_case("binomial_qg6x6_k4")
    tensors = init_plan_tensors(plan, seed=12)
    y = evaluate_plan(plan, as_torch(tensors), torch.from_numpy(g["x"].astype(np.int64)))
    assert np.array_equal(y.numpy(), g["y_f32"])
    _assert_same_plan(image_data((1, 6, 6), "quad-graph", input_layer="binomial", num_input_units=4, sum_product_layer="cp",
                                 num_sum_units=4), plan)
