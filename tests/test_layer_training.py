"""Training THROUGH the layer functions (row b2: what the TorchLayer subclasses of cirkit_amd/cirkit_plugin.py call), end to
end, pinned by the reference's own `loss.backward()`:

* fixtures `tests/golden/*_grads.npz` hold the parameter gradients the reference's autograd produced (make_fixtures.py
  `grads`, `grads_tucker`, `grads_sos`);
* on CPU the oracle's autograd (oracle/torch_oracle.py, op-for-op restatement) reproduces the NEW squared-circuit fixture --
  ComplexLSESumSemiring.apply_reduce (semiring.py:441-476) and ComplexSafeLog (utils.py:22-50) under autograd;
* on the GPU the oracle's interpreter loop (circuits.py:242-278, the gather between layers, the parameter graphs: torch
  autograd, as in the reference) runs with every layer forward replaced by `cirkit_amd.layer_ops` -- HIP forward kernels
  and the hand-written backward kernels behind `torch.autograd.Function`s -- and must give the fixture's gradients.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.plan import Plan  # noqa: E402
from oracle import torch_oracle as oracle  # noqa: E402  (tests may: the oracle is the checker)


def _sos_case():
    plan_c, plan_z = Plan.load(os.path.join(GOLDEN, "sos_4x4_c_k4")), Plan.load(os.path.join(GOLDEN, "sos_4x4_z_k4"))
    with np.load(os.path.join(GOLDEN, "sos_4x4_k4_grads.npz")) as z:
        ref = {k: z[k] for k in z.files}
    return plan_c, plan_z, init_plan_tensors(plan_c), ref


def _sos_loss(plan_c, plan_z, leaves, x):
    c = oracle.evaluate_plan(plan_c, leaves, x, grad=True)
    z = oracle.evaluate_plan(plan_z, leaves, None, grad=True)
    return -(2.0 * c.real - z.real).mean()


def test_oracle_autograd_reproduces_the_squared_circuit_gradients():
    plan_c, plan_z, tensors, ref = _sos_case()
    leaves = {k: torch.from_numpy(np.ascontiguousarray(v)).requires_grad_(True) for k, v in tensors.items()}
    loss = _sos_loss(plan_c, plan_z, leaves, torch.from_numpy(ref["x"].astype(np.int64)))
    loss.backward()
    assert abs(loss.item() - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
    for k in tensors:
        want = ref["g_" + k]
        assert np.abs(leaves[k].grad.numpy() - want).max() <= 2e-5 * max(1e-3, np.abs(want).max()), k


def _hip_layer_forward(sr, l, params, x):
    """oracle._layer_forward with the layer bodies of cirkit_amd/cirkit_plugin.py (same tensors in, same tensors out)."""
    from cirkit_amd import _capi as capi
    from cirkit_amd import layer_ops as ops

    cplx = sr is oracle._CLSE
    t = l.type
    if t == "categorical":
        logits = torch.log(params["probs"]) if "probs" in params else params["logits"]
        return sr.from_lse(ops.categorical_log_likelihood(x, logits))
    if t == "gaussian":
        return sr.from_lse(ops.gaussian_log_likelihood(x, params["mean"], params["stddev"], params.get("log_partition")))
    if t == "embedding":
        return ops.embedding(x, params["weight"], complex_out=cplx)
    if t == "constant":
        return ops.constant_value(params["value"], int(x), log_space=bool(l.config.get("log_space")), complex_out=cplx)
    if t == "hadamard":
        return ops.hadamard(x)
    if t == "kronecker":
        return ops.kronecker(x)
    if t in ("sum", "cpt", "tucker"):
        mode = {"sum": capi.CK_SUM_CAT, "cpt": capi.CK_SUM_PROD, "tucker": capi.CK_SUM_KRON}[t]
        return ops.sum_lse(x, params["weight"], mode)
    if t == "tensordot":
        kj = int(params["weight"].shape[2])
        return ops.tensordot_lse(x, params["weight"], kj, l.num_input_units // kj)
    raise NotImplementedError(t)


def _with_hip_layers(fn):
    stock = oracle._layer_forward
    oracle._layer_forward = _hip_layer_forward
    try:
        return fn()
    finally:
        oracle._layer_forward = stock


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg1_rbt8", "quadgraph_6x6_k4", "pd_gauss_6x6_k4", "quadgraph_6x6_tucker_k4", "quadtree_4x4_kron_k3"])
def test_training_through_the_layer_functions_real(hip_device, name):
    plan = Plan.load(os.path.join(GOLDEN, name))
    tensors = init_plan_tensors(plan)
    with np.load(os.path.join(GOLDEN, name + "_grads.npz")) as z:
        ref = {k: z[k] for k in z.files}
    x = torch.from_numpy(ref["x"].astype(np.float32 if ref["x"].dtype.kind == "f" else np.int64)).to(hip_device)
    leaves = {k: torch.from_numpy(np.ascontiguousarray(v)).to(hip_device).requires_grad_(True) for k, v in tensors.items()}
    loss = _with_hip_layers(lambda: -oracle.evaluate_plan(plan, leaves, x, grad=True).mean())
    loss.backward()
    assert abs(loss.item() - float(ref["loss"])) <= 2e-5 * abs(float(ref["loss"]))
    for k in tensors:
        want = ref["g_" + k]
        got = leaves[k].grad.cpu().numpy()
        assert np.abs(got - want).max() <= 5e-4 * max(1e-3, np.abs(want).max()), (k, float(np.abs(got - want).max()), float(np.abs(want).max()))


@pytest.mark.gpu
def test_training_through_the_layer_functions_squared_circuit(hip_device):
    """loss = -mean(2 Re c(x) - Re Z) of a squared circuit: Embedding, CP-T under complex-lse-sum for c; ConstantValue,
    Hadamard and TensorDot layers (with the reference's kron / einsum parameter graphs, evaluated by torch) for Z."""
    plan_c, plan_z, tensors, ref = _sos_case()
    leaves = {k: torch.from_numpy(np.ascontiguousarray(v)).to(hip_device).requires_grad_(True) for k, v in tensors.items()}
    x = torch.from_numpy(ref["x"].astype(np.int64)).to(hip_device)
    loss = _with_hip_layers(lambda: _sos_loss(plan_c, plan_z, leaves, x))
    loss.backward()
    assert abs(loss.item() - float(ref["loss"])) <= 2e-5 * abs(float(ref["loss"]))
    for k in tensors:
        want = ref["g_" + k]
        got = leaves[k].grad.cpu().numpy()
        assert np.abs(got - want).max() <= 1e-3 * max(1e-3, np.abs(want).max()), (k, float(np.abs(got - want).max()), float(np.abs(want).max()))
