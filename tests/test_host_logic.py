"""Host-side logic that needs no GPU: the layer/parameter operator surface mirrored from the
reference, and the plan analysis behind cross-layer fusion."""
import numpy as np
import pytest
import torch

from conftest import load_case
from cirkit_amd.fusion import find_subtree_groups
from cirkit_amd.layers import (
    HipCategoricalLayer,
    HipCPTLayer,
    HipHadamardLayer,
    HipKroneckerLayer,
    HipSumLayer,
    HipTensorDotLayer,
    layer_from_spec,
)
from cirkit_amd.parameters import HipParameter, TensorStore, _einsum_as_bmm
from cirkit_amd.plan import IDX_NONE, FoldIndex, ParamGraph, ParamNode, resolve_fold_index


def _param(store, name, shape, F, ops=("tensor",)):
    store.set(name, np.zeros((F, *shape), dtype=np.float32))
    nodes = [ParamNode("tensor", F, tuple(shape), {"tensor": name}, [])]
    for op in ops[1:]:
        nodes.append(ParamNode(op, F, tuple(shape), {"dim": len(shape) - 1}, [FoldIndex([len(nodes) - 1], IDX_NONE)]))
    return HipParameter(ParamGraph(nodes, FoldIndex([len(nodes) - 1], IDX_NONE), F, tuple(shape)), store)


def test_layer_surface_mirrors_the_reference():
    store = TensorStore("cpu")
    w = _param(store, "w", (8, 16), 3, ("tensor", "softmax"))
    l = HipSumLayer(8, 8, 2, weight=w, num_folds=3)
    assert dict(l.config) == {"num_input_units": 8, "num_output_units": 8, "arity": 2}
    assert list(l.params) == ["weight"] and l.params["weight"].shape == (8, 16)
    assert l.fold_settings == (("num_input_units", 8), ("num_output_units", 8), ("arity", 2), ("weight", (8, 16)))
    # re-instantiation the way the reference's folding does it (compiler.py:398-406)
    again = type(l)(semiring=l.semiring, **l.config, num_folds=3, **l.params)
    assert again.fold_settings == l.fold_settings
    assert w.softmax_source() is store["w"]
    c = HipCPTLayer(8, 4, 2, weight=_param(store, "w2", (4, 8), 1), num_folds=1)
    assert c._weight_shape == (4, 8)
    h = HipHadamardLayer(8, 3, num_folds=2)
    assert h.num_output_units == 8 and dict(h.config) == {"num_input_units": 8, "arity": 3}
    k = HipKroneckerLayer(4, 2)
    assert k.num_output_units == 16
    t = HipTensorDotLayer(12, 20, weight=_param(store, "w3", (5, 3), 2), num_folds=2)
    assert (t._num_contract_units, t._num_batch_units) == (3, 4)


def test_constructor_errors_match_the_reference_messages():
    store = TensorStore("cpu")
    with pytest.raises(ValueError, match="Expected number of folds 2"):
        HipSumLayer(8, 8, 1, weight=_param(store, "a", (8, 8), 3), num_folds=2)  # inner.py:237-242
    with pytest.raises(ValueError, match="shape"):
        HipSumLayer(8, 8, 2, weight=_param(store, "b", (8, 8), 1), num_folds=1)
    with pytest.raises(ValueError, match="arity should be at least 2"):
        HipHadamardLayer(8, 1)  # inner.py:112-113
    with pytest.raises(ValueError, match="arity must be a positive integer"):
        HipSumLayer(8, 8, 0, weight=_param(store, "c", (8, 0), 1))
    with pytest.raises(ValueError, match="Exactly one between"):
        HipCategoricalLayer(np.arange(4)[:, None], 8, num_categories=3)  # input.py:349-350
    with pytest.raises(ValueError, match="univariate"):
        HipCategoricalLayer(np.zeros((4, 2), dtype=np.int64), 8, num_categories=3, probs=_param(store, "d", (8, 3), 4))
    with pytest.raises(ValueError, match="K_jK_q"):
        HipTensorDotLayer(12, 21, weight=_param(store, "e", (5, 3), 1))  # optimized.py:255-262
    with pytest.raises(ValueError, match="semiring"):
        HipHadamardLayer(8, 2, semiring="sum-product")


@pytest.mark.parametrize("name", ["cfg1_rbt8", "cfg2_qt784", "cfg4_pd784", "cfg5_sos_c_k32", "cfg5_sos_z_k32",
                                  "kat_bernoulli_f0o0", "kat_gaussian_f1o1"])
def test_every_fixture_plan_instantiates(name):
    plan, tensors, _ = load_case(name)
    store = TensorStore("cpu")
    store.update(tensors)
    layers = [layer_from_spec(s, store, plan.semiring) for s in plan.layers]
    for s, l in zip(plan.layers, layers):
        assert (l.num_folds, l.num_output_units) == (s.num_folds, s.num_output_units)
        assert dict(l.config) == s.config
    mixing = [l for l in layers if getattr(l, "_mixing", False)]
    if name == "cfg4_pd784":
        assert len(mixing) == 10  # ten Sum(mix) layers; layer 35 is a dense SumCollapse
        assert not layers[35]._mixing and layers[35].weight.ops[-1] == "matmul"


def test_einsum_patterns_map_to_bmm():
    assert _einsum_as_bmm([[0, 1], [2, 1], [0, 2]], [(4, 9), (5, 9)]) == (False, 4, 5, 9, 0, 1)
    assert _einsum_as_bmm([[0, 1], [1, 2], [0, 2]], [(4, 9), (9, 5)]) == (False, 4, 5, 9, 0, 0)
    assert _einsum_as_bmm([[1, 0], [1, 2], [0, 2]], [(9, 4), (9, 5)]) == (False, 4, 5, 9, 1, 0)
    assert _einsum_as_bmm([[1, 2], [0, 1], [0, 2]], [(9, 5), (4, 9)]) == (True, 4, 5, 9, 0, 0)
    assert _einsum_as_bmm([[0, 1], [0, 1], [0, 1]], [(4, 9), (4, 9)]) is None


def _setup(name):
    plan, tensors, g = load_case(name)
    store = TensorStore("cpu")
    store.update(tensors)
    layers = [layer_from_spec(s, store, plan.semiring) for s in plan.layers]
    folds = [l.num_folds for l in layers]
    children = [None if s.inputs is None else resolve_fold_index(s.inputs, folds) for s in plan.layers]
    out_pairs = resolve_fold_index(plan.output, folds).reshape(-1, 2)
    return plan, tensors, g, layers, children, out_pairs


def test_fusion_group_of_the_quadtree_plan():
    plan, tensors, g, layers, children, out_pairs = _setup("cfg2_qt784")
    (grp,) = find_subtree_groups(plan, layers, children, out_pairs, 4)
    # layer 5 (49 folds) is read by layers 6 AND 7, so it must be materialised: the chain stops there
    assert (grp.input_layer, grp.dense_layer, grp.levels, grp.root) == (0, 1, [2, 3, 4, 5], 5)
    assert grp.virtual == [0, 1, 2, 3, 4]
    assert list(grp.nodes[grp.leaf_off : grp.leaf_off + 16]) == [0, 1, 28, 29, 2, 3, 30, 31, 56, 57, 84, 85, 58, 59, 86, 87]
    for depth in (1, 2, 3):
        (g2,) = find_subtree_groups(plan, layers, children, out_pairs, depth)
        assert g2.depth == depth and g2.root == 1 + depth


def test_fusion_tables_describe_the_same_computation():
    """Walk the node tables on the CPU with the oracle's per-layer arithmetic (binary-counter order
    of ck_fused.hip) and compare with the oracle's layer-wise result for the root layer."""
    from oracle.torch_oracle import _LSE, _layer_forward, as_torch, eval_param, evaluate_plan

    plan, tensors, g, layers, children, out_pairs = _setup("cfg2_qt784")
    (grp,) = find_subtree_groups(plan, layers, children, out_pairs, 3)
    tt = as_torch(tensors)
    x = torch.from_numpy(g["x"].astype(np.int64))[:4]
    _, outs = evaluate_plan(plan, tt, x, return_all=True)
    D, kL = grp.depth, 1 << grp.depth
    P = {i: {pn: eval_param(pg, tt) for pn, pg in plan.layers[i].params.items()} for i in [0, 1] + grp.levels}
    logits = torch.log(P[0]["probs"])  # (F, K, C)
    nodes = grp.nodes

    def sum_step(w, v):  # v: (B, K) -> (B, K), one fold
        m = v.amax(dim=-1, keepdim=True)
        return torch.log(torch.exp(v - m) @ w.T) + m

    for t in (0, 17, 97):
        stack = [None] * D
        cur = None
        for i in range(kL):
            c = int(nodes[grp.leaf_off + t * kL + i])
            d = int(nodes[grp.node_off[0] + t * kL + i])
            cur = logits[c][:, x[:, int(plan.layers[0].scope_idx[c, 0])]].T  # (B, K)
            cur = sum_step(P[1]["weight"][d], cur)
            for l in range(D):
                if ((i >> l) & 1) == 0:
                    stack[l] = cur
                    break
                fold = int(nodes[grp.node_off[l + 1] + t * (kL >> (l + 1)) + (i >> (l + 1))])
                cur = sum_step(P[grp.levels[l]]["weight"][fold], stack[l] + cur)
        want = outs[grp.root][t]
        assert float((cur - want).abs().max()) < 1e-3 * float(want.abs().max())


def test_no_fusion_where_the_pattern_does_not_hold():
    for name in ("cfg1_rbt8", "cfg4_pd784", "cfg5_sos_c_k32"):
        plan, tensors, g, layers, children, out_pairs = _setup(name)
        assert find_subtree_groups(plan, layers, children, out_pairs, 4) == []


def test_tensordot_lists_of_a_squared_circuits_partition_function():
    """`fusion.tensordot_lists` on Z = integral |c|^2 of a squared QuadTree circuit (symbolic/operators.py:39-322 applied to the
    plan of c, cirkit_amd/functional.py): every sum layer of c became a PAIR of TensorDot layers (W, conj W) over one Hadamard
    layer -- the second reads the first fold by fold, the first reads the Hadamard layer fold by fold, nobody else reads either:
    one launch per sum layer, the Hadamard layers read as lists; the output layer is never absorbed."""
    from cirkit_amd.functional import squared_partition_plan
    from cirkit_amd.fusion import tensordot_lists
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    plan_c = image_data((1, 8, 8), "quad-tree-2", input_layer="embedding", num_input_units=4, sum_product_layer="cp-t", num_sum_units=4,
                        sum_weight_activation="none", semiring="complex-lse-sum")
    z = squared_partition_plan(plan_c)
    store = TensorStore("cpu")
    store.update(init_plan_tensors(plan_c))
    layers = [layer_from_spec(s, store, z.semiring) for s in z.layers]
    folds = [l.num_folds for l in layers]
    children = [None if s.inputs is None else resolve_fold_index(s.inputs, folds) for s in z.layers]
    out = {int(p) for p in resolve_fold_index(z.output, folds).reshape(-1, 2)[:, 0]}
    had_of, pair_of = tensordot_lists(layers, children, out)
    td = [i for i, l in enumerate(layers) if isinstance(l, HipTensorDotLayer)]
    had = [i for i, l in enumerate(layers) if isinstance(l, HipHadamardLayer)]
    n_sum = sum(1 for s in plan_c.layers if s.type in ("cpt", "sum"))
    assert len(td) == 2 * n_sum and len(pair_of) == n_sum and len(had_of) == len(had) == n_sum
    for b, a in pair_of.items():  # second over first, first over its Hadamard layer
        assert a in had_of and b not in had_of and a not in out and b > a > had_of[a]
        assert layers[a].num_folds == layers[b].num_folds == layers[had_of[a]].num_folds
    assert set(pair_of) | set(pair_of.values()) == set(td)
    # a layer some other fusion owns, or that holds the output, is left alone
    first = min(pair_of.values())
    h2, p2 = tensordot_lists(layers, children, out, busy={had_of[first]})
    assert first not in h2 and len(h2) == n_sum - 1 and p2 == pair_of
    top = max(pair_of)
    h3, p3 = tensordot_lists(layers, children, out | {pair_of[top]})
    assert top not in p3 and len(p3) == n_sum - 1
