"""Generate the committed parity fixtures by running the REAL reference (build container only).

    PYTHONPATH=/root/repo python tests/golden/make_fixtures.py

Imports april-tools/cirkit from /root/reference (read-only), compiles the BASELINE.json
configurations with ``fold=True, optimize=True``, loads the closed-form parameters of
``cirkit_amd.initializers`` into the reference circuits (in place, through the storage the plan
extraction shares with them), evaluates them, and writes

    tests/golden/<name>.json / .npz      the folded plan (layer list + index arrays)
    tests/golden/<name>_golden.npz       inputs x, reference outputs in fp32 (and fp64 where cheap),
                                         plus, for the small KAT circuits, the literal weights

Nothing here ships reference code: the outputs are data (ints, floats).  The GPU box never runs this.
"""

from __future__ import annotations

import copy
import functools
import os
import sys

import numpy as np
import torch

REF = os.environ.get("CIRKIT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cirkit.symbolic.functional as SF  # noqa: E402
from cirkit.pipeline import PipelineContext  # noqa: E402
from cirkit.templates import data_modalities  # noqa: E402
from cirkit.templates.utils import Parameterization  # noqa: E402

from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.plan import plan_from_torch_circuit, tensor_table  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def _load_closed_form(plan, tensors, seed=0):
    vals = init_plan_tensors(plan, seed=seed)
    for k, t in tensors.items():
        t.copy_(torch.from_numpy(vals[k]).to(t.dtype))


def _save(name, plan, extra):
    plan.name = name
    plan.save(os.path.join(HERE, name))
    np.savez_compressed(os.path.join(HERE, name + "_golden.npz"), **extra)
    print(f"{name}: {len(plan.layers)} layers, {plan.num_params} params, "
          f"fixture arrays {sum(v.nbytes for v in extra.values())} B")


def _fp64_copy(cc):
    c64 = copy.deepcopy(cc)
    return c64.double()


def cfg1():
    sc = data_modalities.tabular_data(
        "random-binary-tree", num_features=8,
        input_layers={"name": "categorical", "args": {"num_categories": 4}},
        num_input_units=4, sum_product_layer="cp", num_sum_units=4)
    cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    plan, tensors = plan_from_torch_circuit(cc)
    _load_closed_form(plan, tensors)
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 4, (32, 8), generator=g)
    y32 = cc(x)
    y64 = _fp64_copy(cc)(x)
    _save("cfg1_rbt8", plan, {"x": x.numpy(), "y_f32": y32.numpy(), "y_f64": y64.numpy()})
    # unfolded / unoptimised compilation of the same circuit with the same weights must agree
    # (reference invariant, tests/backend/torch/test_compile_circuit.py:87-101 parametrisation)


def cfg2():
    sc = data_modalities.image_data(
        (1, 28, 28), "quad-tree-2", input_layer="categorical", num_input_units=32,
        sum_product_layer="cp", num_sum_units=32)
    cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    plan, tensors = plan_from_torch_circuit(cc)
    _load_closed_form(plan, tensors)
    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, 256, (64, 784), generator=g)
    y32 = cc(x)
    y64 = _fp64_copy(cc)(x)
    _save("cfg2_qt784", plan, {"x": x.numpy().astype(np.int16), "y_f32": y32.numpy(), "y_f64": y64.numpy()})


def cfg2_cpt():
    """Same region graph with ``cp-t`` (no dense layer after the inputs) -- SURVEY.md A.2 note."""
    sc = data_modalities.image_data(
        (1, 28, 28), "quad-tree-2", input_layer="categorical", num_input_units=16,
        sum_product_layer="cp-t", num_sum_units=16)
    cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    plan, tensors = plan_from_torch_circuit(cc)
    _load_closed_form(plan, tensors)
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (16, 784), generator=g)
    _save("cfg2t_qt784_cpt16", plan, {"x": x.numpy().astype(np.int16), "y_f32": cc(x).numpy()})


def cfg4():
    sc = data_modalities.image_data(
        (1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64,
        sum_product_layer="cp", num_sum_units=64)
    cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    plan, tensors = plan_from_torch_circuit(cc)
    _load_closed_form(plan, tensors)
    g = torch.Generator().manual_seed(4)
    x = torch.randn((16, 784), generator=g)
    y32 = cc(x)
    y64 = _fp64_copy(cc)(x.double())
    _save("cfg4_pd784", plan, {"x": x.numpy(), "y_f32": y32.numpy(), "y_f64": y64.numpy()})


def cfg5(K=32):
    """Squared (SoS) circuit: c(x) under complex-lse-sum and its partition function Z."""
    par = Parameterization(activation="none", initialization="normal")
    sc = data_modalities.image_data(
        (1, 28, 28), "quad-tree-2", input_layer="embedding", num_input_units=K,
        sum_product_layer="cp-t", num_sum_units=K,
        input_params={"weight": par}, sum_weight_param=par)
    ctx = PipelineContext(backend="torch", semiring="complex-lse-sum", fold=True, optimize=True)
    cc = ctx.compile(sc)
    zc = ctx.compile(SF.integrate(SF.multiply(sc, SF.conjugate(sc))))
    table = tensor_table()
    plan_c, tensors = plan_from_torch_circuit(cc, table=table)
    plan_z, tensors_z = plan_from_torch_circuit(zc, table=table)
    assert set(tensors_z) <= set(tensors), "Z must only point at c's tensors"
    # scale the closed-form values so that K-term sums stay O(1)
    vals = init_plan_tensors(plan_c)
    for k, t in tensors.items():
        t.copy_(torch.from_numpy(vals[k]).to(t.dtype))
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (16, 784), generator=g)
    y = cc(x)
    z = zc()
    _save(f"cfg5_sos_c_k{K}", plan_c, {"x": x.numpy().astype(np.int16), "y_c64": y.numpy()})
    _save(f"cfg5_sos_z_k{K}", plan_z, {"z_c64": z.numpy()})


def sq_categorical():
    """A REAL circuit squared: Categorical inputs, CP-T layers, lse-sum.  Z = integrate(multiply(c, c)) from the
    reference (operators.py:51-63 integrate Categorical, :106-139 Categorical x Categorical), with the closed-form
    parameters; the native functional.squared_partition_plan must reproduce the value."""
    sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", input_layer="categorical", num_input_units=5,
                                    sum_product_layer="cp-t", num_sum_units=5)
    ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
    cc = ctx.compile(sc)
    zc = ctx.compile(SF.integrate(SF.multiply(sc, sc)))
    table = tensor_table()
    plan_c, tensors = plan_from_torch_circuit(cc, table=table)
    _load_closed_form(plan_c, tensors, seed=6)
    g = torch.Generator().manual_seed(6)
    x = torch.randint(0, 256, (8, 16), generator=g)
    _save("sq_cat_qt4x4_k5", plan_c, {"x": x.numpy().astype(np.int16), "y_f32": cc(x).numpy(), "z_f32": zc().numpy(),
                                      "z_f64": _fp64_copy(zc)().numpy()})


def sq_gaussian():
    """A real circuit with Gaussian inputs, squared: Z = integrate(multiply(c, c)) from the reference
    (operators.py:66-77, 142-200; nodes.py:975-988)."""
    sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", input_layer="gaussian", num_input_units=4,
                                    sum_product_layer="cp", num_sum_units=4)
    ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
    cc = ctx.compile(sc)
    zc = ctx.compile(SF.integrate(SF.multiply(sc, sc)))
    table = tensor_table()
    plan_c, tensors = plan_from_torch_circuit(cc, table=table)
    _load_closed_form(plan_c, tensors, seed=8)
    g = torch.Generator().manual_seed(8)
    x = torch.randn((8, 16), generator=g)
    _save("sq_gauss_qt4x4_k4", plan_c, {"x": x.numpy(), "y_f32": cc(x).numpy(), "z_f32": zc().numpy(),
                                        "z_f64": _fp64_copy(zc)().numpy()})


def complex_inputs():
    """Categorical (logits) and Gaussian input layers under complex-lse-sum: the reference maps their real
    log-likelihoods into the complex semiring (layers/input.py:276-278, semiring.py:512-514) and the signed sum
    weights make the activations genuinely complex from the first sum layer on.  c(x) on 12 rows in complex64 and
    complex128."""
    par = Parameterization(activation="none", initialization="normal")
    for name, kw, seed in (
        ("sos_cat_c_qt4x4_k6", dict(input_layer="categorical", num_input_units=6, num_sum_units=6,
                                    input_params={"logits": par}), 21),
        ("sos_gauss_c_qt4x4_k4", dict(input_layer="gaussian", num_input_units=4, num_sum_units=4), 22),
    ):
        sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", sum_product_layer="cp-t", sum_weight_param=par, **kw)
        ctx = PipelineContext(backend="torch", semiring="complex-lse-sum", fold=True, optimize=True)
        cc = ctx.compile(sc)
        plan, tensors = plan_from_torch_circuit(cc)
        _load_closed_form(plan, tensors)
        g = torch.Generator().manual_seed(seed)
        if kw["input_layer"] == "categorical":
            x = torch.randint(0, 256, (12, 16), generator=g)
            xs, x64 = x.numpy().astype(np.int16), x
        else:
            x = torch.randn((12, 16), generator=g)
            xs, x64 = x.numpy(), x.double()
        y = cc(x)
        y128 = _fp64_copy(cc)(x64)
        _save(name, plan, {"x": xs, "y_c64": y.numpy(), "y_c128": y128.numpy()})


def binomial():
    """Binomial input layers (image_data(..., input_layer="binomial"): total_count 255, probs = sigmoid(tensor)):
    plan + outputs of the reference on 12 rows, closed-form parameters."""
    sc = data_modalities.image_data((1, 6, 6), "quad-graph", input_layer="binomial", num_input_units=4,
                                    sum_product_layer="cp", num_sum_units=4)
    ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
    cc = ctx.compile(sc)
    plan, tensors = plan_from_torch_circuit(cc)
    _load_closed_form(plan, tensors, seed=12)
    g = torch.Generator().manual_seed(12)
    x = torch.randint(0, 256, (12, 36), generator=g)
    x[0, :] = 0
    x[1, :] = 255  # the ends of the support
    _save("binomial_qg6x6_k4", plan, {"x": x.numpy().astype(np.int16), "y_f32": cc(x).numpy(), "y_f64": _fp64_copy(cc)(x).numpy()})


def kats():
    """The reference's own known-answer circuits (tests/symbolic/test_utils.py:293-503), compiled by
    the reference with fold+optimize under lse-sum; literal weights stored (they are tiny)."""
    sys.path.insert(0, REF)
    from tests.symbolic.test_utils import (
        build_monotonic_bivariate_gaussian_hadamard_dense_pc,
        build_monotonic_structured_categorical_cpt_pc,
    )
    import itertools

    for fold, optimize in itertools.product([False, True], [False, True]):
        tag = f"f{int(fold)}o{int(optimize)}"
        ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=fold, optimize=optimize)
        sc, gt, zgt = build_monotonic_structured_categorical_cpt_pc(return_ground_truth=True)
        cc = ctx.compile(sc)
        plan, tensors = plan_from_torch_circuit(cc)
        worlds = torch.tensor(list(itertools.product([0, 1], repeat=5)))
        y = cc(worlds)
        extra = {"x": worlds.numpy(), "y_f32": y.numpy(),
                 "kat_x": np.array(list(gt["evi"].keys())),
                 "kat_y": np.array(list(gt["evi"].values())),
                 "kat_z": np.array(zgt)}
        extra.update({"w_" + k: v.numpy() for k, v in tensors.items()})
        _save(f"kat_bernoulli_{tag}", plan, extra)

        sc, gt, zgt = build_monotonic_bivariate_gaussian_hadamard_dense_pc(return_ground_truth=True)
        cc = ctx.compile(sc)
        plan, tensors = plan_from_torch_circuit(cc)
        xs = torch.tensor([[0.3, 1.2], [0.0, 0.0], [-1.5, 2.5], [4.0, -3.0]])
        y = cc(xs)
        extra = {"x": xs.numpy(), "y_f32": y.numpy(),
                 "kat_x": np.array(list(gt["evi"].keys())),
                 "kat_y": np.array(list(gt["evi"].values())),
                 "kat_z": np.array(zgt)}
        extra.update({"w_" + k: v.numpy() for k, v in tensors.items()})
        _save(f"kat_gaussian_{tag}", plan, extra)


def tucker():
    """Tucker sum-product layers (Kronecker -> dense fused into TorchTuckerLayer): small QuadTree."""
    sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", input_layer="categorical", num_input_units=6,
                                    sum_product_layer="tucker", num_sum_units=6)
    cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    plan, tensors = plan_from_torch_circuit(cc)
    _load_closed_form(plan, tensors)
    g = torch.Generator().manual_seed(6)
    x = torch.randint(0, 256, (24, 16), generator=g)
    _save("tucker_qt16_k6", plan, {"x": x.numpy().astype(np.int16), "y_f32": cc(x).numpy(),
                                   "y_f64": _fp64_copy(cc)(x).numpy()})


def templates_extra():
    """Golden outputs for template features beyond the BASELINE configs: a Tucker block of arity 4
    (quad-tree-4), a quad-graph (two partitionings per region: mixing layers + collapsed sums) and a
    tabular circuit with per-feature input families (categorical and Gaussian columns in one batch)."""
    ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
    g = torch.Generator().manual_seed(9)
    cases = [
        ("tucker4_qt16_k3", data_modalities.image_data((1, 4, 4), "quad-tree-4", input_layer="categorical", num_input_units=3,
                                                       sum_product_layer="tucker", num_sum_units=3),
         torch.randint(0, 256, (24, 16), generator=g)),
        ("quadgraph_6x6_k4", data_modalities.image_data((1, 6, 6), "quad-graph", input_layer="categorical", num_input_units=4,
                                                        sum_product_layer="cp", num_sum_units=4),
         torch.randint(0, 256, (24, 36), generator=g)),
    ]
    cases.append(("pd_gauss_6x6_k4", data_modalities.image_data((1, 6, 6), "poon-domingos", input_layer="gaussian", num_input_units=4,
                                                       sum_product_layer="cp", num_sum_units=4),
                  torch.randn(24, 36, generator=g)))
    xm = torch.randn(24, 6, generator=g)
    xm[:, 0::2] = torch.randint(0, 3, (24, 3), generator=g).float()
    cases.append(("rbt6_perfeature_k2", data_modalities.tabular_data(
        "random-binary-tree", num_features=6,
        input_layers=[{"name": "categorical", "args": {"num_categories": 3}}, {"name": "gaussian", "args": {}}] * 3,
        num_input_units=2, sum_product_layer="cp", num_sum_units=2), xm))
    for name, sc, x in cases:
        cc = ctx.compile(sc)
        plan, tensors = plan_from_torch_circuit(cc)
        _load_closed_form(plan, tensors)
        xs = x.numpy().astype(np.float32 if x.is_floating_point() else np.int16)
        _save(name, plan, {"x": xs, "y_f32": cc(x).numpy(), "y_f64": _fp64_copy(cc)(x).numpy()})


def grads():
    """Parameter gradients of loss = -mean(log p) from the reference's own autograd (training path,
    notebooks/learning-a-circuit.ipynb cell 18).  cfg1: every gradient; cfg2 (B = 16): per-tensor
    norm, sum and the first 16 entries."""
    torch.set_grad_enabled(True)
    try:
        for name, build, xgen in [
            ("cfg1_rbt8", lambda: data_modalities.tabular_data(
                "random-binary-tree", num_features=8, input_layers={"name": "categorical", "args": {"num_categories": 4}},
                num_input_units=4, sum_product_layer="cp", num_sum_units=4),
             lambda g: torch.randint(0, 4, (32, 8), generator=g)),
            ("cfg2_qt784", lambda: data_modalities.image_data(
                (1, 28, 28), "quad-tree-2", input_layer="categorical", num_input_units=32,
                sum_product_layer="cp", num_sum_units=32),
             lambda g: torch.randint(0, 256, (16, 784), generator=g)),
            # mixing layers, a collapsed Sum -> Sum pair (MatMul weight), Gaussian inputs (scaled-sigmoid stddev)
            ("quadgraph_6x6_k4", lambda: data_modalities.image_data(
                (1, 6, 6), "quad-graph", input_layer="categorical", num_input_units=4,
                sum_product_layer="cp", num_sum_units=4),
             lambda g: torch.randint(0, 256, (24, 36), generator=g)),
            ("pd_gauss_6x6_k4", lambda: data_modalities.image_data(
                (1, 6, 6), "poon-domingos", input_layer="gaussian", num_input_units=4,
                sum_product_layer="cp", num_sum_units=4),
             lambda g: torch.randn(24, 36, generator=g)),
        ]:
            cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(build())
            plan, tensors = plan_from_torch_circuit(cc)
            with torch.no_grad():
                _load_closed_form(plan, tensors)
            g = torch.Generator().manual_seed(11)
            x = xgen(g)
            loss = -cc(x).mean()
            loss.backward()
            by_ptr = {p.data_ptr(): p for p in cc.parameters()}
            extra = {"x": x.numpy().astype(np.float32 if x.is_floating_point() else np.int16), "loss": np.array(loss.item())}
            for k, t in tensors.items():
                gr = by_ptr[t.data_ptr()].grad
                if not name.startswith("cfg2"):
                    extra["g_" + k] = gr.numpy()
                else:
                    extra["gnorm_" + k] = np.array(gr.norm().item())
                    extra["gsum_" + k] = np.array(gr.double().sum().item())
                    extra["ghead_" + k] = gr.reshape(-1)[:16].numpy()
            np.savez_compressed(os.path.join(HERE, name + "_grads.npz"), **extra)
            print(name, "grads:", {k: float(np.linalg.norm(v)) for k, v in extra.items() if k.startswith("g_")} or "summaries")
    finally:
        torch.set_grad_enabled(False)


def grads_k64():
    """Round 5: the reference's own `loss.backward()` on 64-unit CP circuits of the two kinds the job form of the training step
    (cirkit_amd/train_jobs.py) covers -- a QuadGraph circuit with Categorical leaves (the model of notebooks/learning-a-circuit.ipynb
    cells 4 / 16 / 18, at 6x6 pixels) and a Poon-Domingos circuit with Gaussian leaves (BASELINE config 4 at 6x6): the plans, the
    loss, and per parameter tensor its gradient's norm, sum and first 32 entries (the tensors themselves come from
    cirkit_amd.initializers, closed form)."""
    torch.set_grad_enabled(True)
    try:
        for name, build, xgen in [
            ("quadgraph_6x6_k64", lambda: data_modalities.image_data(
                (1, 6, 6), "quad-graph", input_layer="categorical", num_input_units=64, sum_product_layer="cp", num_sum_units=64),
             lambda g: torch.randint(0, 256, (48, 36), generator=g)),
            ("pd_gauss_6x6_k64", lambda: data_modalities.image_data(
                (1, 6, 6), "poon-domingos", input_layer="gaussian", num_input_units=64, sum_product_layer="cp", num_sum_units=64),
             lambda g: torch.randn(48, 36, generator=g)),
        ]:
            cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(build())
            plan, tensors = plan_from_torch_circuit(cc)
            with torch.no_grad():
                _load_closed_form(plan, tensors)
            g = torch.Generator().manual_seed(17)
            x = xgen(g)
            y = cc(x)
            loss = -y.mean()
            loss.backward()
            by_ptr = {p.data_ptr(): p for p in cc.parameters()}
            extra = {"x": x.numpy().astype(np.float32 if x.is_floating_point() else np.int16), "loss": np.array(loss.item()),
                     "y_f32": y.detach().numpy()}
            for k, t in tensors.items():
                gr = by_ptr[t.data_ptr()].grad
                extra["gnorm_" + k] = np.array(gr.norm().item())
                extra["gsum_" + k] = np.array(gr.double().sum().item())
                extra["ghead_" + k] = gr.reshape(-1)[:32].numpy()
            plan.name = name
            plan.save(os.path.join(HERE, name))
            np.savez_compressed(os.path.join(HERE, name + "_grads.npz"), **extra)
            print(name, "loss", float(loss), "tensors", len(tensors), "layers", len(plan.layers))
    finally:
        torch.set_grad_enabled(False)


def grads_tucker():
    """Round 3: parameter gradients of the reference's autograd through Tucker layers (optimized.py:89-103) and through
    stand-alone Kronecker + dense sum layers (inner.py:178-187; `optimize=False` keeps them unfused) -- the plans, a forward
    fixture and every gradient."""
    torch.set_grad_enabled(True)
    try:
        for name, sc, optimize, xgen in [
            ("quadgraph_6x6_tucker_k4", data_modalities.image_data(
                (1, 6, 6), "quad-graph", input_layer="categorical", num_input_units=4,
                sum_product_layer="tucker", num_sum_units=4), True,
             lambda g: torch.randint(0, 256, (24, 36), generator=g)),
            ("quadtree_4x4_kron_k3", data_modalities.image_data(
                (1, 4, 4), "quad-tree-2", input_layer="categorical", num_input_units=3,
                sum_product_layer="tucker", num_sum_units=3), False,
             lambda g: torch.randint(0, 256, (20, 16), generator=g)),
        ]:
            cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=optimize).compile(sc)
            plan, tensors = plan_from_torch_circuit(cc)
            with torch.no_grad():
                _load_closed_form(plan, tensors)
            g = torch.Generator().manual_seed(13)
            x = xgen(g)
            with torch.no_grad():
                _save(name, plan, {"x": x.numpy().astype(np.int16), "y_f32": cc(x).numpy(), "y_f64": _fp64_copy(cc)(x).numpy()})
            loss = -cc(x).mean()
            loss.backward()
            by_ptr = {p.data_ptr(): p for p in cc.parameters()}
            extra = {"x": x.numpy().astype(np.int16), "loss": np.array(loss.item())}
            for k, t in tensors.items():
                extra["g_" + k] = by_ptr[t.data_ptr()].grad.numpy()
            np.savez_compressed(os.path.join(HERE, name + "_grads.npz"), **extra)
            print(name, [l.type for l in plan.layers], "grads:", {k: float(np.linalg.norm(v)) for k, v in extra.items() if k.startswith("g_")})
    finally:
        torch.set_grad_enabled(False)


def grads_param_nodes():
    """Round 4: the reference's autograd through parameter nodes the default templates do not produce -- a softmax along an
    INNER axis (TorchSoftmaxParameter with dim != last, nodes.py:764-772: sum weights normalised over the OUTPUT units) and a
    sigmoid activation (nodes.py:656-699) -- plan, forward fixture and every gradient."""
    torch.set_grad_enabled(True)
    try:
        for name, wpar in [
            ("quadtree_4x4_softmax0_k4", Parameterization(activation="softmax", initialization="normal", activation_kwargs={"axis": 0})),
            ("quadtree_4x4_sigmoid_k4", Parameterization(activation="sigmoid", initialization="normal")),
            # round 6: the two remaining activations of templates/utils.py name_to_parameter_activation (:189-194):
            # TorchSoftplusParameter (nodes.py:731-739) and TorchClampParameter with vmin = 1e-18 (nodes.py:702-728) -- with the
            # closed-form pseudo-normal parameters about half the entries sit ON the clamp, where the gradient is zero
            ("quadtree_4x4_softplus_k4", Parameterization(activation="softplus", initialization="normal")),
            ("quadtree_4x4_posclamp_k4", Parameterization(activation="positive-clamp", initialization="normal")),
        ]:
            if os.environ.get("ONLY_NEW") and os.path.exists(os.path.join(HERE, name + "_grads.npz")):
                continue
            sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", input_layer="categorical", num_input_units=4,
                                            sum_product_layer="cp", num_sum_units=4, sum_weight_param=wpar)
            cc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
            plan, tensors = plan_from_torch_circuit(cc)
            with torch.no_grad():
                _load_closed_form(plan, tensors)
            g = torch.Generator().manual_seed(17)
            x = torch.randint(0, 256, (20, 16), generator=g)
            with torch.no_grad():
                _save(name, plan, {"x": x.numpy().astype(np.int16), "y_f32": cc(x).numpy(), "y_f64": _fp64_copy(cc)(x).numpy()})
            loss = -cc(x).mean()
            loss.backward()
            by_ptr = {p.data_ptr(): p for p in cc.parameters()}
            extra = {"x": x.numpy().astype(np.int16), "loss": np.array(loss.item())}
            for k, t in tensors.items():
                extra["g_" + k] = by_ptr[t.data_ptr()].grad.numpy()
            np.savez_compressed(os.path.join(HERE, name + "_grads.npz"), **extra)
            print(name, [(l.type, [n.op + str(n.config.get("dim", "")) for pg in l.params.values() for n in pg.nodes]) for l in plan.layers][:3],
                  "grads:", {k: float(np.linalg.norm(v)) for k, v in extra.items() if k.startswith("g_")})
    finally:
        torch.set_grad_enabled(False)


def grads_sos():
    """Round 3: the reference's autograd through the COMPLEX semiring (ComplexLSESumSemiring.apply_reduce, semiring.py:441-476;
    ComplexSafeLog, utils.py:22-50): a small squared circuit c (Embedding, CP-T, signed weights) and its partition function
    Z = integral |c|^2, loss = -mean(2 Re c(x) - Re Z) -- the negative log-likelihood of the squared model -- and the gradient
    of every tensor."""
    torch.set_grad_enabled(True)
    try:
        par = Parameterization(activation="none", initialization="normal")
        sc = data_modalities.image_data(
            (1, 4, 4), "quad-tree-2", input_layer="embedding", num_input_units=4,
            sum_product_layer="cp-t", num_sum_units=4, input_params={"weight": par}, sum_weight_param=par)
        ctx = PipelineContext(backend="torch", semiring="complex-lse-sum", fold=True, optimize=True)
        cc = ctx.compile(sc)
        zc = ctx.compile(SF.integrate(SF.multiply(sc, SF.conjugate(sc))))
        table = tensor_table()
        plan_c, tensors = plan_from_torch_circuit(cc, table=table)
        plan_z, tensors_z = plan_from_torch_circuit(zc, table=table)
        assert set(tensors_z) <= set(tensors)
        with torch.no_grad():
            vals = init_plan_tensors(plan_c)
            for k, t in tensors.items():
                t.copy_(torch.from_numpy(vals[k]).to(t.dtype))
        g = torch.Generator().manual_seed(17)
        x = torch.randint(0, 256, (12, 16), generator=g)
        with torch.no_grad():
            _save("sos_4x4_c_k4", plan_c, {"x": x.numpy().astype(np.int16), "y_c64": cc(x).numpy()})
            _save("sos_4x4_z_k4", plan_z, {"z_c64": zc().numpy()})
        loss = -(2.0 * cc(x).real - zc().real).mean()
        loss.backward()
        by_ptr = {p.data_ptr(): p for p in cc.parameters()}
        extra = {"x": x.numpy().astype(np.int16), "loss": np.array(loss.item())}
        for k, t in tensors.items():
            extra["g_" + k] = by_ptr[t.data_ptr()].grad.numpy()
        np.savez_compressed(os.path.join(HERE, "sos_4x4_k4_grads.npz"), **extra)
        print("sos_4x4", [l.type for l in plan_c.layers], [l.type for l in plan_z.layers],
              "loss", loss.item(), {k: float(np.linalg.norm(v)) for k, v in extra.items() if k.startswith("g_")})
    finally:
        torch.set_grad_enabled(False)


def grads_sq_categorical():
    """Round 4: the reference's autograd through a REAL squared circuit (the circuit of `sq_categorical`): loss =
    -mean(2 c(x) - Z) with Z = integrate(multiply(c, c)) -- ConstantValue layers whose values are logs of Gram matrices of
    softmaxed probability rows, Hadamard and TensorDot layers under lse-sum -- and the gradient of every tensor."""
    torch.set_grad_enabled(True)
    try:
        sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", input_layer="categorical", num_input_units=5,
                                        sum_product_layer="cp-t", num_sum_units=5)
        ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
        cc = ctx.compile(sc)
        zc = ctx.compile(SF.integrate(SF.multiply(sc, sc)))
        table = tensor_table()
        plan_c, tensors = plan_from_torch_circuit(cc, table=table)
        with torch.no_grad():
            _load_closed_form(plan_c, tensors, seed=6)
        g = torch.Generator().manual_seed(6)
        x = torch.randint(0, 256, (8, 16), generator=g)
        loss = -(2.0 * cc(x) - zc()).mean()
        loss.backward()
        by_ptr = {p.data_ptr(): p for p in cc.parameters()}
        extra = {"x": x.numpy().astype(np.int16), "loss": np.array(loss.item())}
        for k, t in tensors.items():
            extra["g_" + k] = by_ptr[t.data_ptr()].grad.numpy()
        np.savez_compressed(os.path.join(HERE, "sq_cat_qt4x4_k5_grads.npz"), **extra)
        print("sq_cat grads: loss", loss.item(), {k: float(np.linalg.norm(v)) for k, v in extra.items() if k.startswith("g_")})
    finally:
        torch.set_grad_enabled(False)


def grads_sq_gaussian():
    """Round 4: the same for Gaussian inputs (the circuit of `sq_gaussian`): the constant layers of Z hold the closed-form log
    integrals of products of two Gaussian units (nodes.py:975-988) over pointer / scaled-sigmoid graphs."""
    torch.set_grad_enabled(True)
    try:
        sc = data_modalities.image_data((1, 4, 4), "quad-tree-2", input_layer="gaussian", num_input_units=4,
                                        sum_product_layer="cp", num_sum_units=4)
        ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
        cc = ctx.compile(sc)
        zc = ctx.compile(SF.integrate(SF.multiply(sc, sc)))
        table = tensor_table()
        plan_c, tensors = plan_from_torch_circuit(cc, table=table)
        with torch.no_grad():
            _load_closed_form(plan_c, tensors, seed=8)
        g = torch.Generator().manual_seed(8)
        x = torch.randn((8, 16), generator=g)
        loss = -(2.0 * cc(x) - zc()).mean()
        loss.backward()
        by_ptr = {p.data_ptr(): p for p in cc.parameters()}
        extra = {"x": x.numpy(), "loss": np.array(loss.item())}
        for k, t in tensors.items():
            extra["g_" + k] = by_ptr[t.data_ptr()].grad.numpy()
        np.savez_compressed(os.path.join(HERE, "sq_gauss_qt4x4_k4_grads.npz"), **extra)
        print("sq_gauss grads: loss", loss.item(), {k: float(np.linalg.norm(v)) for k, v in extra.items() if k.startswith("g_")})
    finally:
        torch.set_grad_enabled(False)


def marginals():
    """Marginal queries through the reference's IntegrateQuery (cirkit/backend/torch/queries.py):
    the KAT circuits (reference ground truth: mar (1,0,1,1,.) = 16.845, Z = 318; mar (0.3,.) =
    23.528960785605985, Z = 44) and random per-row masks on configs 1 and 2."""
    from cirkit.backend.torch.queries import IntegrateQuery
    from tests.symbolic.test_utils import (
        build_monotonic_bivariate_gaussian_hadamard_dense_pc,
        build_monotonic_structured_categorical_cpt_pc,
    )

    ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
    sc, gt, zgt = build_monotonic_structured_categorical_cpt_pc(return_ground_truth=True)
    cc = ctx.compile(sc)
    q = IntegrateQuery(cc)
    x = torch.tensor([[1, 0, 1, 1, 0], [1, 0, 1, 1, 1], [0, 1, 0, 0, 1], [1, 1, 1, 0, 0]])
    mask = torch.tensor([[0, 0, 0, 0, 1], [1, 1, 1, 1, 1], [1, 0, 1, 0, 0], [0, 0, 0, 0, 0]], dtype=torch.bool)
    y = q(x, integrate_vars=mask)
    np.savez_compressed(os.path.join(HERE, "kat_bernoulli_f1o1_marg.npz"), x=x.numpy(), mask=mask.numpy(), y=y.numpy(),
                        kat_mar=np.array(list(gt["mar"].values())), kat_z=np.array(zgt))
    print("bernoulli marginals", torch.exp(y).flatten().tolist(), gt["mar"], zgt)

    sc, gt, zgt = build_monotonic_bivariate_gaussian_hadamard_dense_pc(return_ground_truth=True)
    cc = ctx.compile(sc)
    q = IntegrateQuery(cc)
    x = torch.tensor([[0.3, 1.2], [0.3, 1.2], [0.3, 1.2], [-1.0, 2.0]])
    mask = torch.tensor([[0, 1], [1, 1], [0, 0], [1, 0]], dtype=torch.bool)
    y = q(x, integrate_vars=mask)
    np.savez_compressed(os.path.join(HERE, "kat_gaussian_f1o1_marg.npz"), x=x.numpy(), mask=mask.numpy(), y=y.numpy(),
                        kat_mar=np.array(list(gt["mar"].values())), kat_z=np.array(zgt))
    print("gaussian marginals", torch.exp(y).flatten().tolist(), gt["mar"], zgt)

    for name, build, xgen, B in [
        ("cfg1_rbt8", lambda: data_modalities.tabular_data(
            "random-binary-tree", num_features=8, input_layers={"name": "categorical", "args": {"num_categories": 4}},
            num_input_units=4, sum_product_layer="cp", num_sum_units=4),
         lambda g, B: torch.randint(0, 4, (B, 8), generator=g), 32),
        ("cfg2_qt784", lambda: data_modalities.image_data(
            (1, 28, 28), "quad-tree-2", input_layer="categorical", num_input_units=32,
            sum_product_layer="cp", num_sum_units=32),
         lambda g, B: torch.randint(0, 256, (B, 784), generator=g), 8),
    ]:
        cc = ctx.compile(build())
        plan, tensors = plan_from_torch_circuit(cc)
        _load_closed_form(plan, tensors)
        g = torch.Generator().manual_seed(21)
        x = xgen(g, B)
        mask = torch.rand(x.shape, generator=g) < 0.3
        mask[0] = False
        mask[1] = True
        y = IntegrateQuery(cc)(x, integrate_vars=mask)
        np.savez_compressed(os.path.join(HERE, name + "_marg.npz"), x=x.numpy().astype(np.int16), mask=mask.numpy(), y=y.numpy())
        print(name, "marginals", y.flatten()[:4].tolist())


def plans_only():
    """Plan-only fixtures (no outputs) that pin the native plan builders of cirkit_amd/templates.py on
    awkward shapes: odd borders, single rows, quad-tree-4, deeper random trees."""
    ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
    cases = []
    for (h, w), rgname, sp in [((5, 7), "quad-tree-2", "cp"), ((3, 3), "quad-tree-2", "cp-t"), ((1, 6), "quad-tree-2", "cp"),
                               ((6, 1), "quad-tree-2", "cp-t"), ((9, 4), "quad-tree-4", "cp"), ((2, 2), "quad-tree-2", "cp"),
                               ((7, 7), "quad-tree-4", "cp-t")]:
        sc = data_modalities.image_data((1, h, w), rgname, input_layer="categorical", num_input_units=3,
                                        sum_product_layer=sp, num_sum_units=3)
        cases.append((f"plan_{rgname.replace('-', '')}_{h}x{w}_{sp.replace('-', '')}", sc))
    for n, depth, sp in [(13, None, "cp"), (6, 2, "cp-t"), (21, 3, "cp")]:
        kw = {} if depth is None else {"region_graph_args": {"depth": depth}}
        try:
            sc = data_modalities.tabular_data("random-binary-tree", num_features=n,
                                              input_layers={"name": "categorical", "args": {"num_categories": 3}},
                                              num_input_units=2, sum_product_layer=sp, num_sum_units=2, **kw)
        except TypeError:
            if depth is not None:
                continue
            raise
        cases.append((f"plan_rbt{n}_d{depth}_{sp.replace('-', '')}", sc))
    # region graphs with several partitionings per region (mixing layers, collapsed sums), Tucker
    # blocks, multi-channel pixels (factorised inputs), shallow random trees, per-feature inputs
    img = dict(input_layer="categorical", num_input_units=3, num_sum_units=3)
    for shape, rgname, sp, extra in [
        ((1, 4, 4), "quad-graph", "cp", {}), ((1, 5, 7), "quad-graph", "cp-t", {}), ((1, 6, 6), "quad-graph", "tucker", {}),
        ((3, 4, 4), "quad-graph", "cp", {}), ((1, 8, 8), "quad-graph", "cp", {"use_mixing_weights": False}),
        ((1, 4, 4), "quad-graph", "cp", {"num_classes": 3}), ((1, 4, 4), "quad-tree-4", "tucker", {}),
        ((2, 3, 5), "quad-tree-2", "cp-t", {}), ((1, 8, 8), "poon-domingos", "cp", {}),
        ((1, 12, 10), "poon-domingos", "cp-t", {}), ((1, 20, 20), "poon-domingos", "tucker", {}),
        ((2, 17, 9), "poon-domingos", "cp", {}), ((1, 12, 13), "random-binary-tree", "cp", {}),
    ]:
        sc = data_modalities.image_data(shape, rgname, sum_product_layer=sp, **img, **extra)
        tag = "".join(f"_{k}{v}" for k, v in extra.items()).replace("use_mixing_weights", "mix").replace("num_classes", "nc")
        cases.append((f"plan_{rgname.replace('-', '')}_{'x'.join(map(str, shape))}_{sp.replace('-', '')}{tag}", sc))

    from cirkit.symbolic.parameters import mixing_weight_factory
    from cirkit.templates.region_graph import PoonDomingos, RandomBinaryTree
    from cirkit.templates.utils import Parameterization, name_to_input_layer_factory, parameterization_to_factory

    def from_rg(rg, sp, input_name="categorical", **input_kwargs):
        swf = parameterization_to_factory(Parameterization(activation="softmax", initialization="normal"))
        return rg.build_circuit(
            input_factory=name_to_input_layer_factory(input_name, **input_kwargs), sum_product=sp, sum_weight_factory=swf,
            nary_sum_weight_factory=functools.partial(mixing_weight_factory, param_factory=swf),
            num_input_units=2, num_sum_units=2, num_classes=1, factorize_multivariate=True)

    cases.append(("plan_pd_1x12x12_delta6-3_cp", from_rg(PoonDomingos((1, 12, 12), delta=[6, 3]), "cp", num_categories=5)))
    cases.append(("plan_pd_1x9x6_delta2_depth2_cp", from_rg(PoonDomingos((1, 9, 6), delta=2, max_depth=2), "cp", num_categories=5)))
    cases.append(("plan_rbt11_d2_cp", from_rg(RandomBinaryTree(11, depth=2), "cp", num_categories=3)))
    cases.append(("plan_rbt19_d3_cpt", from_rg(RandomBinaryTree(19, depth=3, seed=7), "cp-t", num_categories=3)))
    from cirkit.templates.region_graph import FullyFactorized, LinearTree

    cases.append(("plan_ff5_r1_cp", from_rg(FullyFactorized(5), "cp", num_categories=3)))
    cases.append(("plan_ff4_r3_cp", from_rg(FullyFactorized(4, num_repetitions=3), "cp", num_categories=3)))
    cases.append(("plan_lt6_r1_cp", from_rg(LinearTree(6), "cp", num_categories=3)))
    cases.append(("plan_lt5_r2_rand3_cpt", from_rg(LinearTree(5, num_repetitions=2, randomize=True, seed=3), "cp-t", num_categories=3)))
    cases.append(("plan_lt4_order2031_cp", from_rg(LinearTree(4, ordering=[2, 0, 3, 1]), "cp", num_categories=3)))
    cases.append(("plan_rbt6_perfeature_cp", data_modalities.tabular_data(
        "random-binary-tree", num_features=6,
        input_layers=[{"name": "categorical", "args": {"num_categories": 3}}, {"name": "gaussian", "args": {}}] * 3,
        num_input_units=2, sum_product_layer="cp", num_sum_units=2)))
    for name, sc in cases:
        plan, _ = plan_from_torch_circuit(ctx.compile(sc))
        plan.name = name
        plan.save(os.path.join(HERE, name))
        print(name, len(plan.layers), "layers")


def _clt_datasets():
    """Seeded tabular datasets with a dependence structure worth learning (a shuffled noisy chain /
    a few latent factors), small enough to commit."""
    rng = np.random.default_rng(11)
    # categorical: a noisy Markov chain over a shuffled variable order, 4 categories
    n, d, c = 400, 9, 4
    order = rng.permutation(d)
    cat = np.zeros((n, d), dtype=np.int64)
    cat[:, order[0]] = rng.integers(0, c, n)
    for a, b in zip(order[:-1], order[1:]):
        flip = rng.random(n) < 0.35
        cat[:, b] = np.where(flip, rng.integers(0, c, n), cat[:, a])
    # gaussian: 7 features driven by 2 latent factors with different loadings
    z = rng.standard_normal((300, 2))
    load = rng.standard_normal((2, 7))
    gau = (z @ load + 0.6 * rng.standard_normal((300, 7))).astype(np.float32)
    # mixed: categorical columns 0, 2, 4 (3 categories) and Gaussian columns 1, 3, 5 that follow them
    m = 350
    mix = np.zeros((m, 6), dtype=np.float32)
    base = rng.integers(0, 3, m)
    for j in range(6):
        if j % 2 == 0:
            base = np.where(rng.random(m) < 0.3, rng.integers(0, 3, m), base)
            mix[:, j] = base
        else:
            mix[:, j] = base + 0.5 * rng.standard_normal(m)
    return cat, gau, mix


def chow_liu():
    """Plans of HCLT circuits: the structure is LEARNED from data by the reference
    (templates/region_graph/algorithms/chow_liu.py) -- the data, the learned tree (list of predecessors)
    and the compiled plan are committed; cirkit_amd.templates must learn the same tree and emit the same plan."""
    from cirkit.templates.region_graph.algorithms.chow_liu import ChowLiuTree

    ctx = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)
    cat, gau, mix = _clt_datasets()
    cases = [
        ("plan_clt_cat9_cp", torch.from_numpy(cat), {"name": "categorical", "args": {"num_categories": 4}}, "cp"),
        ("plan_clt_gauss7_cpt", torch.from_numpy(gau), {"name": "gaussian", "args": {}}, "cp-t"),
        ("plan_clt_mixed6_cp", torch.from_numpy(mix),
         [{"name": "categorical", "args": {"num_categories": 3}}, {"name": "gaussian", "args": {}}] * 3, "cp"),
    ]
    for name, data, inputs, sp in cases:
        sc = data_modalities.tabular_data("chow-liu-tree", data=data, input_layers=inputs, num_input_units=3,
                                          sum_product_layer=sp, num_sum_units=3)
        itype = inputs["name"] if isinstance(inputs, dict) else [i["name"] for i in inputs]
        ncat = inputs["args"]["num_categories"] if isinstance(inputs, dict) and inputs["name"] == "categorical" else None
        tree = ChowLiuTree(data, itype, num_categories=ncat, as_region_graph=False)
        plan, _ = plan_from_torch_circuit(ctx.compile(sc))
        plan.name = name
        plan.save(os.path.join(HERE, name))
        np.savez_compressed(os.path.join(HERE, name + "_data.npz"), data=data.numpy(), tree=np.asarray(tree, dtype=np.int64))
        print(name, len(plan.layers), "layers; tree", list(map(int, tree)))


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg2", "cfg2_cpt", "cfg4", "cfg5", "kats", "tucker", "plans_only", "grads", "grads_k64", "grads_tucker", "grads_param_nodes", "grads_sos", "grads_sq_categorical", "grads_sq_gaussian", "marginals", "templates_extra"]
    for w in which:
        globals()[w]()
