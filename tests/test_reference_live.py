"""Live comparison with the real reference -- only where it is importable (the build container: /root/reference).
Compiles circuits with april-tools/cirkit itself and checks, on the spot, that (1) the native plan builder emits the
reference's folded plan, (2) the oracle reproduces the reference's forward bit for bit with the same parameters, and
(3) the natively built partition function of a squared circuit equals the reference's `integrate(multiply(c, c))`.
The committed fixtures pin a fixed set of circuits; this sweeps further combinations every time the CPU suite runs
here.  Skipped wherever the reference is absent (e.g. on the GPU box)."""
import os
import sys

import numpy as np
import pytest
import torch

REF = os.environ.get("CIRKIT_REFERENCE", "/root/reference")
if not os.path.isdir(os.path.join(REF, "cirkit")):
    pytest.skip("the reference checkout is not available here", allow_module_level=True)
sys.path.insert(0, REF)
try:
    import cirkit.symbolic.functional as SF  # noqa: E402
    from cirkit.pipeline import PipelineContext  # noqa: E402
    from cirkit.templates import data_modalities  # noqa: E402
except Exception as e:  # pragma: no cover - depends on the environment
    pytest.skip(f"the reference does not import here: {e}", allow_module_level=True)

from cirkit_amd.functional import squared_partition_plan  # noqa: E402
from cirkit_amd.plan import plan_from_torch_circuit, tensor_table  # noqa: E402
from cirkit_amd.templates import image_data, tabular_data  # noqa: E402
from oracle.torch_oracle import evaluate_plan  # noqa: E402
from test_templates import _assert_same_plan  # noqa: E402

CASES = [
    ((1, 5, 4), "quad-graph", "categorical", "cp", 3, 1, True),
    ((1, 6, 6), "quad-tree-4", "categorical", "tucker", 2, 1, True),
    ((2, 4, 4), "quad-tree-2", "gaussian", "cp-t", 3, 1, True),
    ((1, 8, 8), "poon-domingos", "gaussian", "cp", 2, 1, True),
    ((1, 4, 6), "random-binary-tree", "binomial", "cp", 3, 1, True),
    ((1, 6, 6), "quad-graph", "categorical", "cp", 3, 4, False),
    ((1, 7, 5), "poon-domingos", "binomial", "cp-t", 2, 1, True),
]


@pytest.mark.parametrize("shape,rg,inp,sp,k,ncls,mix", CASES, ids=[f"{c[1]}-{c[2]}-{c[3]}" for c in CASES])
def test_plan_and_forward_match_the_live_reference(shape, rg, inp, sp, k, ncls, mix):
    torch.manual_seed(0)
    kw = dict(input_layer=inp, num_input_units=k, sum_product_layer=sp, num_sum_units=k, num_classes=ncls, use_mixing_weights=mix)
    sc = data_modalities.image_data(shape, rg, **kw)
    cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    plan, tensors = plan_from_torch_circuit(cc)
    _assert_same_plan(image_data(shape, rg, **kw), plan)
    d = int(np.prod(shape))
    g = torch.Generator().manual_seed(1)
    x = torch.randn((6, d), generator=g) if inp == "gaussian" else torch.randint(0, 256, (6, d), generator=g)
    with torch.no_grad():
        want = cc(x)
        got = evaluate_plan(plan, {n: t.detach() for n, t in tensors.items()}, x)
    assert torch.equal(got, want)


@pytest.mark.parametrize("activation,native", [("softplus", "softplus"), ("positive-clamp", "positive-clamp"), ("sigmoid", "sigmoid")])
def test_sum_weight_activations_match_the_live_reference(activation, native):
    """The sum-weight activations templates/utils.py name_to_parameter_activation names beside softmax (:185-194): the natively
    built plan is the reference's (TorchSoftplusParameter / TorchClampParameter with vmin = 1e-18 / TorchSigmoidParameter nodes)
    and the oracle reproduces the reference's forward bit for bit."""
    from cirkit.templates.utils import Parameterization

    torch.manual_seed(0)
    sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", input_layer="categorical", num_input_units=3, sum_product_layer="cp",
                                    num_sum_units=3, sum_weight_param=Parameterization(activation=activation, initialization="normal"))
    cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    plan, tensors = plan_from_torch_circuit(cc)
    _assert_same_plan(image_data((1, 4, 4), "quad-tree-2", input_layer="categorical", num_input_units=3, sum_product_layer="cp",
                                 num_sum_units=3, sum_weight_activation=native), plan)
    x = torch.randint(0, 256, (6, 16), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = cc(x)
        got = evaluate_plan(plan, {n: t.detach() for n, t in tensors.items()}, x)
    assert torch.equal(got, want)


@pytest.mark.parametrize("inp", ["categorical", "gaussian"])
def test_squared_partition_function_matches_the_live_reference(inp):
    torch.manual_seed(0)
    sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", input_layer=inp, num_input_units=3, sum_product_layer="cp-t",
                                    num_sum_units=3)
    ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
    cc = ctx.compile(sc)
    zc = ctx.compile(SF.integrate(SF.multiply(sc, sc)))
    plan, tensors = plan_from_torch_circuit(cc, table=tensor_table())
    with torch.no_grad():
        want = float(zc().reshape(-1)[0])
        got = float(evaluate_plan(squared_partition_plan(plan), {n: t.detach() for n, t in tensors.items()}, None).reshape(-1)[0])
    assert abs(got - want) <= 1e-5 * abs(want)


def test_complex_squared_circuit_matches_the_live_reference():
    """Embedding inputs, unconstrained weights, complex-lse-sum: c(x) through the oracle and Z built natively."""
    from cirkit.templates.utils import Parameterization

    torch.manual_seed(0)
    par = Parameterization(activation="none", initialization="normal")
    sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", input_layer="embedding", num_input_units=3, sum_product_layer="cp-t",
                                    num_sum_units=3, input_params={"weight": par}, sum_weight_param=par)
    ctx = PipelineContext(backend="torch", semiring="complex-lse-sum", fold=True, optimize=True)
    cc = ctx.compile(sc)
    zc = ctx.compile(SF.integrate(SF.multiply(sc, SF.conjugate(sc))))
    plan, tensors = plan_from_torch_circuit(cc, table=tensor_table())
    _assert_same_plan(image_data((1, 4, 4), "quad-tree-2", input_layer="embedding", num_input_units=3, sum_product_layer="cp-t",
                                 num_sum_units=3, sum_weight_activation="none", semiring="complex-lse-sum"), plan)
    tens = {n: t.detach() for n, t in tensors.items()}
    x = torch.randint(0, 256, (5, 16), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        assert torch.equal(evaluate_plan(plan, tens, x), cc(x))
        want = zc().reshape(-1)[0]
        got = evaluate_plan(squared_partition_plan(plan), tens, None).reshape(-1)[0]
    assert abs(float(got.real) - float(want.real)) <= 1e-5 * abs(float(want.real))


@pytest.mark.parametrize("cls,kwargs,shapes,op,config", [
    ("TorchHadamardParameter", {}, [(3, 5), (3, 5)], "hadamard", {}),
    ("TorchKroneckerParameter", {}, [(2, 3), (4, 5)], "kronecker", {}),
    ("TorchOuterProductParameter", {"dim": 1}, [(3, 2), (3, 4)], "outer_product", {"dim": 1}),
    ("TorchReduceSumParameter", {"dim": 0}, [(4, 2, 3)], "reduce_sum", {"dim": 0}),
    ("TorchSumParameter", {}, [(4, 3), (4, 3)], "sum", {}),
    ("TorchReduceProductParameter", {"dim": 1}, [(3, 6)], "reduce_prod", {"dim": 1}),
    ("TorchReduceLSEParameter", {"dim": 0}, [(4, 2, 3)], "reduce_lse", {"dim": 0}),
    ("TorchOuterSumParameter", {"dim": 1}, [(3, 2), (3, 4)], "outer_sum", {"dim": 1}),
    ("TorchIndexParameter", {"indices": [2, 0, 2, 4], "dim": 0}, [(5, 3)], "index", {"indices": [2, 0, 2, 4], "dim": 0}),
    ("TorchGaussianProductMean", {}, [(4, 1), (4, 1), (3, 1), (3, 1)], "gaussian_product_mean", {}),
    ("TorchGaussianProductStddev", {}, [(4, 1), (3, 1)], "gaussian_product_stddev", {}),
    ("TorchClampParameter", {"vmin": 1e-18}, [(5, 7)], "clamp", {"vmin": 1e-18}),
    ("TorchClampParameter", {"vmin": -0.3, "vmax": 0.4}, [(5, 7)], "clamp", {"vmin": -0.3, "vmax": 0.4}),
    ("TorchSoftplusParameter", {}, [(5, 7)], "softplus", {}),
])
def test_oracle_parameter_nodes_equal_the_reference_modules(cls, kwargs, shapes, op, config):
    """The oracle's restatement of the parameter nodes added in round 6 (nodes.py:491-612, 702-751) against the reference's own
    modules on the same operands: bit for bit, and the plan extraction maps each class to its op."""
    from cirkit.backend.torch.parameters import nodes as ref_nodes

    from cirkit_amd.plan import IDX_NONE, PARAM_OPS, FoldIndex, ParamGraph, ParamNode
    from oracle.torch_oracle import eval_param

    assert PARAM_OPS[cls] == op
    F = 3
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn((F, *s), generator=g) for s in shapes]
    if "Gaussian" in cls:
        xs = [x.abs() + 0.3 for x in xs]
    mod = getattr(ref_nodes, cls)(*shapes, num_folds=F, **kwargs)
    with torch.no_grad():
        want = mod(*xs)
    nodes = [ParamNode("tensor", F, tuple(s), {"tensor": f"t{i}"}, []) for i, s in enumerate(shapes)]
    nodes.append(ParamNode(op, F, tuple(mod.shape), dict(config), [FoldIndex([i], IDX_NONE) for i in range(len(shapes))]))
    pg = ParamGraph(nodes, FoldIndex([len(nodes) - 1], IDX_NONE), F, tuple(mod.shape))
    got = eval_param(pg, {f"t{i}": x for i, x in enumerate(xs)})
    assert got.shape == want.shape and torch.equal(got, want)
