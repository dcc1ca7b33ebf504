"""Row b2: the layer compilation rules of cirkit_amd/cirkit_plugin.py inside the REAL reference
compiler.  Needs april-tools/cirkit importable (the build container has it under /root/reference; skipped elsewhere).
No forward is called here -- there is no GPU in the build container and the subclasses have no CPU path; their forward
bodies (cirkit_amd/layer_ops.py) are tested on the GPU box in tests/test_gpu_layer_ops.py."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
if os.path.isdir(os.path.join(REF, "cirkit")) and REF not in sys.path:
    sys.path.insert(0, REF)
cirkit = pytest.importorskip("cirkit")

from cirkit.pipeline import PipelineContext  # noqa: E402
from cirkit.templates import data_modalities, utils  # noqa: E402


def _registered(ctx):
    """`plugin.register` takes the context's compiler from the CALLER (cirkit exposes no accessor for it; a user who built the
    context owns that choice -- `HipLayersContext` needs nothing of the kind)."""
    import cirkit_amd.cirkit_plugin as plugin

    return plugin.register(ctx, ctx._compiler)


def _symbolic(kind):
    if kind == "qt_cat_cp":
        return data_modalities.image_data((1, 8, 8), region_graph="quad-tree-2", input_layer="categorical", num_input_units=8,
                                          sum_product_layer="cp", num_sum_units=8,
                                          sum_weight_param=utils.Parameterization(activation="softmax", initialization="normal"))
    if kind == "qg_tucker":
        return data_modalities.image_data((1, 6, 6), region_graph="quad-graph", input_layer="categorical", num_input_units=4,
                                          sum_product_layer="tucker", num_sum_units=4,
                                          sum_weight_param=utils.Parameterization(activation="softmax", initialization="normal"))
    return data_modalities.image_data((1, 8, 8), region_graph="poon-domingos", input_layer="gaussian", num_input_units=4,
                                      sum_product_layer="cp", num_sum_units=4,
                                      sum_weight_param=utils.Parameterization(activation="softmax", initialization="normal"))


@pytest.mark.parametrize("kind", ["qt_cat_cp", "qg_tucker", "pd_gauss_cp"])
@pytest.mark.parametrize("fold,optimize", [(False, False), (True, False), (False, True), (True, True)])
def test_registered_rules_survive_optimisation_and_folding(kind, fold, optimize):
    import cirkit_amd.cirkit_plugin as plugin

    sc = _symbolic(kind)
    torch.manual_seed(0)
    stock = PipelineContext(backend="torch", semiring="lse-sum", fold=fold, optimize=optimize).compile(sc)
    torch.manual_seed(0)
    ctx = _registered(PipelineContext(backend="torch", semiring="lse-sum", fold=fold, optimize=optimize))
    cc = ctx.compile(sc)
    a, b = list(stock.layers), list(cc.layers)
    assert len(a) == len(b)
    hip = set(plugin.HIP_LAYER_CLASSES.values())
    for la, lb in zip(a, b):
        # the same circuit: every layer is the HIP SUBCLASS of the stock compile's class, with the same configuration,
        # fold count and parameter shapes (so fusion and folding treated the subclasses exactly like their bases)
        assert type(lb) in hip, type(lb)
        assert plugin.HIP_LAYER_CLASSES[type(la)] is type(lb)
        assert isinstance(lb, type(la))
        assert dict(la.config) == dict(lb.config) and la.num_folds == lb.num_folds
        assert {n: (p.num_folds, tuple(p.shape)) for n, p in la.params.items()} == \
               {n: (p.num_folds, tuple(p.shape)) for n, p in lb.params.items()}
        assert type(lb).forward is not type(la).forward
    # same wiring, and the parameter values the two compilations drew are the same (same seed, same graphs)
    for (ea, eb) in zip(stock.address_book, cc.address_book):
        assert ea.in_module_ids == eb.in_module_ids
    pa = [p.detach() for p in stock.parameters()]
    pb = [p.detach() for p in cc.parameters()]
    assert len(pa) == len(pb) and all(torch.equal(x, y) for x, y in zip(pa, pb))


def test_plan_extraction_accepts_the_plugin_circuit():
    """b4 on top of b2: the plan extracted from a circuit compiled WITH the plugin is the plan of the stock compile (the
    extractor recognises layers by their reference base class)."""
    import cirkit_amd.cirkit_plugin as plugin
    from cirkit_amd.plan import plan_from_torch_circuit

    sc = _symbolic("qt_cat_cp")
    stock = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    cc = _registered(PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)).compile(sc)
    pa, _ = plan_from_torch_circuit(stock)
    pb, _ = plan_from_torch_circuit(cc)
    assert [(l.type, l.num_folds, l.arity, l.num_input_units, l.num_output_units) for l in pa.layers] == \
           [(l.type, l.num_folds, l.arity, l.num_input_units, l.num_output_units) for l in pb.layers]


def test_forward_without_a_rocm_device_fails_loudly():
    import cirkit_amd.cirkit_plugin as plugin
    from cirkit_amd._capi import HipExtensionError

    sc = _symbolic("qt_cat_cp")
    cc = _registered(PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True)).compile(sc)
    with pytest.raises(HipExtensionError):
        cc(torch.randint(0, 256, (4, 64)))
    with pytest.raises(NotImplementedError):
        bad = _registered(PipelineContext(backend="torch", semiring="sum-product", fold=True, optimize=True)).compile(sc)
        bad(torch.randint(0, 256, (4, 64)))


def test_hip_layers_context_and_isolation_of_other_contexts():
    """`HipLayersContext` (no access to private reference state) compiles HIP subclasses; a stock context created AFTER it
    in the same process still compiles stock layers, although the reference shares one rule table between contexts."""
    import cirkit_amd.cirkit_plugin as plugin

    sc = _symbolic("qt_cat_cp")
    cc = plugin.HipLayersContext(semiring="lse-sum", fold=True, optimize=True).compile(sc)
    hip = set(plugin.HIP_LAYER_CLASSES.values())
    assert all(type(l) in hip for l in cc.layers)
    stock = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    assert not any(type(l) in hip for l in stock.layers)
    assert [plugin.HIP_LAYER_CLASSES[type(a)] for a in stock.layers] == [type(b) for b in cc.layers]


def test_shared_storage_entry_points_reevaluate_parameters_at_the_start(monkeypatch):
    """`to_hip(tc)` and `HipPipelineContext.compile(TorchCircuit)` share parameter storage with the reference circuit, so they
    ask for `params_at_end=False` (a write through `p.data` is invisible to `TensorStore.state()`); the native-plan entry keeps
    `HipCircuit`'s own default.  (What the option does is tested on the GPU: test_shared_storage_sees_writes_through_data.)"""
    import cirkit_amd.integration as integ
    import cirkit_amd.pipeline as pipe

    seen = []

    def fake(plan, tensors, **kw):
        seen.append(kw)
        return kw

    monkeypatch.setattr(integ, "HipCircuit", fake)
    monkeypatch.setattr(pipe, "HipCircuit", fake)
    sc = _symbolic("qt_cat_cp")
    tc = PipelineContext(backend="torch", semiring="lse-sum", fold=True, optimize=True).compile(sc)
    integ.to_hip(tc)
    pipe.HipPipelineContext().compile(tc)
    assert [kw["params_at_end"] for kw in seen] == [False, False] and all(kw["pad_units"] is False for kw in seen)
    from cirkit_amd.plan import plan_from_torch_circuit

    plan, tensors = plan_from_torch_circuit(tc)
    pipe.HipPipelineContext().compile(plan, tensors)
    assert "params_at_end" not in seen[-1]
