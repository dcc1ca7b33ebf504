"""Plan-level circuit operators (cirkit_amd/functional.py): the partition function of a squared
circuit built natively from the folded plan of c must reproduce the reference's
``integrate(multiply(c, conjugate(c)))`` (committed output of BASELINE config 5) and the defining
identity  sum_x |c(x)|^2 = Z  (the reference's own invariant, tests/backend/torch/
test_compile_circuit.py:27-51 of the reference)."""
import itertools

import numpy as np
import pytest
import torch

from conftest import load_case
from cirkit_amd.functional import conjugate_plan, squared_partition_plan
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import InputSpec, build_plan, quad_tree, random_binary_tree


def _tiny(sum_product="cp-t", states=3, k=3):
    plan = build_plan(quad_tree(2, 3), input_layer=InputSpec("embedding", states), sum_product=sum_product,
                      num_input_units=k, num_sum_units=k, sum_activation="none", semiring="complex-lse-sum")
    tensors = {n: 0.5 * v for n, v in init_plan_tensors(plan, seed=5).items()}
    worlds = torch.tensor(list(itertools.product(range(states), repeat=6)), dtype=torch.int64)
    return plan, tensors, worlds


def test_native_partition_function_reproduces_reference_output():
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan_c, tensors, _ = load_case("cfg5_sos_c_k32")
    _, _, gz = load_case("cfg5_sos_z_k32")
    z_plan = squared_partition_plan(plan_c)
    assert z_plan.num_variables == 0 and z_plan.layers[0].type == "constant"
    assert [l.type for l in z_plan.layers[1:4]] == ["hadamard", "tensordot", "tensordot"]
    # parameters are shared with c, never copied
    assert all(n.op != "tensor" for l in z_plan.layers for pg in l.params.values() for n in pg.nodes)
    z = evaluate_plan(z_plan, as_torch(tensors), None)
    assert np.array_equal(z.numpy(), gz["z_c64"])


@pytest.mark.parametrize("sum_product", ["cp-t", "cp"])
def test_partition_function_is_the_sum_over_all_worlds(sum_product):
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, worlds = _tiny(sum_product)
    tt = as_torch(tensors)
    c = evaluate_plan(plan, tt, worlds).to(torch.complex128).reshape(-1)
    z = evaluate_plan(squared_partition_plan(plan), tt, None).reshape(-1)[0]
    total = torch.logsumexp(2 * c.real, dim=0)
    assert abs(float(z.real) - float(total)) < 1e-4 * max(1.0, abs(float(total)))
    assert abs(np.remainder(float(z.imag) + np.pi, 2 * np.pi) - np.pi) < 1e-3  # Z is real and positive


def test_conjugate_plan_conjugates_the_output():
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, worlds = _tiny()
    tt = as_torch(tensors)
    c = evaluate_plan(plan, tt, worlds[:32])
    cc = evaluate_plan(conjugate_plan(plan), tt, worlds[:32])
    assert torch.allclose(cc.real, c.real, rtol=1e-6, atol=1e-6)
    d = torch.remainder(cc.imag + c.imag + np.pi, 2 * np.pi) - np.pi
    assert float(d.abs().max()) < 1e-4


def test_unsupported_layers_are_refused():
    from cirkit_amd.templates import image_data

    plan = image_data((1, 4, 4), "quad-tree-2", num_input_units=2, sum_product_layer="tucker", num_sum_units=2)
    with pytest.raises(NotImplementedError):
        squared_partition_plan(plan)


@pytest.mark.gpu
def test_native_partition_function_on_the_gpu(hip_device):
    from cirkit_amd import HipCircuit

    plan_c, tensors, gc = load_case("cfg5_sos_c_k32")
    _, _, gz = load_case("cfg5_sos_z_k32")
    hz = HipCircuit(squared_partition_plan(plan_c), tensors, device=hip_device)
    z = hz().cpu().numpy()
    assert z.shape == gz["z_c64"].shape
    assert abs(z.real - gz["z_c64"].real).max() <= 1e-4 * abs(gz["z_c64"].real).max()
    # log p(x) = 2 Re c(x) - Re Z of the first committed rows
    hc = HipCircuit(plan_c, tensors, device=hip_device)
    x = torch.from_numpy(gc["x"].astype(np.int64)).to(hip_device)
    lp = 2 * hc(x).cpu().numpy().real - z.real
    ref = 2 * gc["y_c64"].real - gz["z_c64"].real
    assert np.allclose(lp, ref, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("sum_product", ["cp-t", "cp"])
def test_partition_function_is_the_sum_over_all_worlds_gpu(hip_device, sum_product):
    from cirkit_amd import HipCircuit

    plan, tensors, worlds = _tiny(sum_product)
    c = HipCircuit(plan, tensors, device=hip_device)(worlds.to(hip_device)).cpu().to(torch.complex128).reshape(-1)
    z = HipCircuit(squared_partition_plan(plan), tensors, device=hip_device)().cpu().reshape(-1)[0]
    total = float(torch.logsumexp(2 * c.real, dim=0))
    assert abs(float(z.real) - total) < 1e-4 * max(1.0, abs(total))


def _sq_cat():
    plan, _, g = load_case("sq_cat_qt4x4_k5")
    return plan, init_plan_tensors(plan, seed=6), g


def test_squared_categorical_circuit_reproduces_reference_partition_function():
    """A real circuit with Categorical inputs, squared: Z = integrate(multiply(c, c)) of the reference
    (tests/golden/make_fixtures.py:sq_categorical) from the natively built plan, and sum_x c(x)^2 = Z by
    enumeration on a circuit small enough to enumerate."""
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, g = _sq_cat()
    assert np.allclose(evaluate_plan(plan, as_torch(tensors), torch.from_numpy(g["x"].astype(np.int64))).numpy(), g["y_f32"], rtol=1e-6)
    z = evaluate_plan(squared_partition_plan(plan), as_torch(tensors), None)
    assert abs(float(z.reshape(-1)[0]) - float(g["z_f64"].reshape(-1)[0])) <= 1e-5 * abs(float(g["z_f64"].reshape(-1)[0]))
    small = build_plan(quad_tree(2, 2), input_layer=InputSpec("categorical", 4), sum_product="cp", num_input_units=3, num_sum_units=3)
    t = init_plan_tensors(small, seed=3)
    worlds = torch.tensor(list(itertools.product(range(4), repeat=4)), dtype=torch.int64)
    ll = evaluate_plan(small, as_torch(t), worlds).double().reshape(-1)
    zs = evaluate_plan(squared_partition_plan(small), as_torch(t), None)
    assert abs(float(zs.reshape(-1)[0]) - float(torch.logsumexp(2 * ll, 0))) <= 1e-5


@pytest.mark.gpu
def test_squared_categorical_partition_function_on_the_gpu(hip_device):
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = _sq_cat()
    hc = HipCircuit(plan, tensors, device=hip_device, pad_units=False)  # Z shares c's (unpadded) parameters
    hz = HipCircuit(squared_partition_plan(plan), hc.store, device=hip_device)
    z = float(hz().cpu().reshape(-1)[0])
    ref = float(g["z_f64"].reshape(-1)[0])
    assert abs(z - ref) <= 1e-4 * abs(ref)


def test_squared_gaussian_circuit_reproduces_reference_partition_function():
    """Gaussian inputs: the integral of the product of two Gaussian units in closed form (nodes.py:975-988)."""
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, _, g = load_case("sq_gauss_qt4x4_k4")
    tensors = init_plan_tensors(plan, seed=8)
    assert np.allclose(evaluate_plan(plan, as_torch(tensors), torch.from_numpy(g["x"])).numpy(), g["y_f32"], rtol=1e-6)
    z = evaluate_plan(squared_partition_plan(plan), as_torch(tensors), None)
    ref = float(g["z_f64"].reshape(-1)[0])
    assert abs(float(z.reshape(-1)[0]) - ref) <= 1e-5 * abs(ref)


@pytest.mark.gpu
def test_squared_gaussian_partition_function_on_the_gpu(hip_device):
    from cirkit_amd.circuit import HipCircuit

    plan, _, g = load_case("sq_gauss_qt4x4_k4")
    tensors = init_plan_tensors(plan, seed=8)
    hc = HipCircuit(plan, tensors, device=hip_device, pad_units=False)
    hz = HipCircuit(squared_partition_plan(plan), hc.store, device=hip_device)
    ref = float(g["z_f64"].reshape(-1)[0])
    assert abs(float(hz().cpu().reshape(-1)[0]) - ref) <= 1e-4 * abs(ref)


@pytest.mark.gpu
def test_partition_function_of_a_padded_circuit(hip_device):
    """Unit counts padded to 32: squaring the PADDED plan (`hc.plan`) over the padded parameter store gives the same Z
    (padded units carry weight 0 everywhere)."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = _sq_cat()
    hc = HipCircuit(plan, tensors, device=hip_device)  # K = 5 -> 32
    assert hc._pad_info is not None
    hz = HipCircuit(squared_partition_plan(hc.plan), hc.store, device=hip_device)
    ref = float(g["z_f64"].reshape(-1)[0])
    assert abs(float(hz().cpu().reshape(-1)[0]) - ref) <= 1e-4 * abs(ref)
