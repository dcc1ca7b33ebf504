"""Plan-level circuit operators: conjugation and the partition function of a squared circuit.

SURVEY.md section 8 (f4).  The reference derives the normaliser of a sum-of-squares model
``p(x) = |c(x)|^2 / Z`` symbolically: ``Z = integrate(multiply(c, conjugate(c)))``
(cirkit/symbolic/functional.py:161-258 integrate, :259-593 multiply, :594-680 conjugate), one rule per
layer pair (cirkit/symbolic/operators.py:39-49 integrate Embedding, :80-103 Embedding x Embedding,
:225-231 Hadamard x Hadamard, :260-270 Sum x Sum with a Kronecker weight, :287-290 / :319-322
conjugation), then compiles the resulting circuit, where the Kronecker-weighted sum is split into
two TensorDot layers (optimization/layers.py:90-127, 282-422).

`squared_partition_plan` applies the same rules directly to the FOLDED plan of ``c``: every folded
layer of ``c`` yields the folded layer(s) of the product circuit with the same fold order, and the
product's parameters are pointers (nodes.py:223-279) to the tensors of ``c`` -- no parameter is
copied or permuted.  The reference reaches an equivalent plan with its folds in another order
(tests/golden/cfg5_sos_z_k32); the value of Z is the same, which tests/test_functional.py pins
against the reference's committed output.
"""

from __future__ import annotations

import numpy as np

from .plan import IDX_ARRAY, IDX_NONE, FoldIndex, LayerSpec, ParamGraph, ParamNode, Plan, resolve_fold_index
from .templates import _fold_index


def _identity(F: int) -> FoldIndex:
    return FoldIndex([0], IDX_ARRAY, np.arange(F, dtype=np.int64))


def _pointer_graph(pg: ParamGraph, *, conjugate: bool) -> ParamGraph:
    """The parameter graph of ``pg`` re-expressed on pointers to its tensors (``Parameter.ref()``,
    symbolic/parameters.py), optionally followed by a complex conjugation."""
    nodes = []
    for n in pg.nodes:
        if n.op == "tensor":
            nodes.append(ParamNode("pointer", n.num_folds, tuple(n.shape), {"tensor": n.config["tensor"], "fold_idx": None}, []))
        else:
            nodes.append(ParamNode(n.op, n.num_folds, tuple(n.shape), dict(n.config),
                                   [FoldIndex(list(fi.ids), fi.kind, None if fi.array is None else fi.array.copy()) for fi in n.inputs]))
    out = FoldIndex(list(pg.output.ids), pg.output.kind, None if pg.output.array is None else pg.output.array.copy())
    if conjugate:
        if out.kind != IDX_ARRAY or len(out.ids) != 1 or not np.array_equal(out.array, np.arange(pg.num_folds)):
            raise NotImplementedError("conjugating a parameter graph whose output re-orders folds")
        nodes.append(ParamNode("conj", pg.num_folds, tuple(pg.shape), {}, [FoldIndex([out.ids[0]], IDX_NONE)]))
        out = FoldIndex([len(nodes) - 1], IDX_ARRAY, np.arange(pg.num_folds, dtype=np.int64))
    return ParamGraph(nodes, out, pg.num_folds, tuple(pg.shape))


def conjugate_plan(plan: Plan) -> Plan:
    """``conjugate(c)``: same structure, every complex parameter conjugated (Embedding weights and
    sum weights; operators.py:287-322), parameters shared with ``c`` through pointers."""
    layers = []
    for l in plan.layers:
        if l.type not in ("embedding", "sum", "cpt", "tucker", "hadamard", "kronecker"):
            raise NotImplementedError(f"conjugation of {l.type!r} layers")
        params = {pn: _pointer_graph(pg, conjugate=True) for pn, pg in l.params.items()}
        layers.append(LayerSpec(l.type, l.num_folds, l.arity, l.num_input_units, l.num_output_units, dict(l.config),
                                params, l.inputs, l.scope_idx))
    return Plan(plan.semiring, plan.num_variables, layers, plan.output, dict(plan.tensors), plan.name + "*")


def _embedding_square_integral(l: LayerSpec, conjugate: bool) -> LayerSpec:
    """sum_s w[k, s] conj(w)[l, s] for every unit pair (k, l): the product of two Embedding layers
    integrated over their variable is a constant layer with K*K units."""
    K = l.num_output_units
    F = l.num_folds
    w = _pointer_graph(l.params["weight"], conjugate=False)
    nodes = list(w.nodes)
    if w.output.kind != IDX_ARRAY or len(w.output.ids) != 1 or not np.array_equal(w.output.array, np.arange(F)):
        raise NotImplementedError("embedding weight graphs whose output re-orders folds")
    a = w.output.ids[0]
    b = a
    if conjugate:
        nodes.append(ParamNode("conj", F, tuple(w.shape), {}, [FoldIndex([a], IDX_NONE)]))
        b = len(nodes) - 1
    nodes.append(ParamNode("einsum", F, (K, K), {"einsum": [[0, 1], [2, 1], [0, 2]]}, [FoldIndex([a], IDX_NONE), FoldIndex([b], IDX_NONE)]))
    nodes.append(ParamNode("flatten", F, (K * K,), {"start_dim": 0, "end_dim": 1}, [FoldIndex([len(nodes) - 1], IDX_NONE)]))
    value = ParamGraph(nodes, FoldIndex([len(nodes) - 1], IDX_ARRAY, np.arange(F, dtype=np.int64)), F, (K * K,))
    return LayerSpec("constant", F, 1, 0, K * K, {"num_output_units": K * K, "log_space": False}, {"value": value})


def _categorical_square_integral(l: LayerSpec) -> LayerSpec:
    """log sum_x p_k(x) p_l(x) for every unit pair (k, l): the product of two Categorical layers (logits added
    unit pair by unit pair, operators.py:106-139) integrated over their variable (the log-sum-exp of those logits
    over the categories, operators.py:51-63) is a constant layer in log space with K*K units.  Written on the
    probabilities: log of the (K, K) Gram matrix of the rows of `probs` (of exp(logits) for a logits layer)."""
    K, F = l.num_output_units, l.num_folds
    src = "probs" if "probs" in l.params else "logits"
    g = _pointer_graph(l.params[src], conjugate=False)
    nodes = list(g.nodes)
    if g.output.kind != IDX_ARRAY or len(g.output.ids) != 1 or not np.array_equal(g.output.array, np.arange(F)):
        raise NotImplementedError("categorical parameter graphs whose output re-orders folds")
    a = g.output.ids[0]
    if src == "logits":
        nodes.append(ParamNode("exp", F, tuple(g.shape), {}, [FoldIndex([a], IDX_NONE)]))
        a = len(nodes) - 1
    nodes.append(ParamNode("einsum", F, (K, K), {"einsum": [[0, 1], [2, 1], [0, 2]]}, [FoldIndex([a], IDX_NONE), FoldIndex([a], IDX_NONE)]))
    nodes.append(ParamNode("log", F, (K, K), {}, [FoldIndex([len(nodes) - 1], IDX_NONE)]))
    nodes.append(ParamNode("flatten", F, (K * K,), {"start_dim": 0, "end_dim": 1}, [FoldIndex([len(nodes) - 1], IDX_NONE)]))
    value = ParamGraph(nodes, FoldIndex([len(nodes) - 1], IDX_ARRAY, np.arange(F, dtype=np.int64)), F, (K * K,))
    return LayerSpec("constant", F, 1, 0, K * K, {"num_output_units": K * K, "log_space": True}, {"value": value})


def _gaussian_square_integral(l: LayerSpec) -> LayerSpec:
    """log of the integral of N(x; m_k, s_k) N(x; m_l, s_l) dx for every unit pair (k, l): the log-partition of the
    product of two Gaussian layers (operators.py:142-200, nodes.py:975-988), which integrating that layer turns
    into a constant layer in log space (operators.py:66-77)."""
    if "log_partition" in l.params:
        raise NotImplementedError("squaring Gaussian layers that carry a log-partition")
    K, F = l.num_output_units, l.num_folds
    nodes: list[ParamNode] = []
    heads = []
    for name in ("mean", "stddev"):
        g = _pointer_graph(l.params[name], conjugate=False)
        if g.output.kind != IDX_ARRAY or len(g.output.ids) != 1 or not np.array_equal(g.output.array, np.arange(F)):
            raise NotImplementedError("Gaussian parameter graphs whose output re-orders folds")
        off = len(nodes)
        for n in g.nodes:  # the two graphs share one node list: shift the operand ids of the second
            nodes.append(ParamNode(n.op, n.num_folds, tuple(n.shape), dict(n.config),
                                   [FoldIndex([i + off for i in fi.ids], fi.kind, None if fi.array is None else fi.array.copy())
                                    for fi in n.inputs]))
        heads.append(g.output.ids[0] + off)
    m, sd = heads
    nodes.append(ParamNode("gaussian_product_log_partition", F, (K * K,), {},
                           [FoldIndex([m], IDX_NONE), FoldIndex([sd], IDX_NONE), FoldIndex([m], IDX_NONE), FoldIndex([sd], IDX_NONE)]))
    value = ParamGraph(nodes, FoldIndex([len(nodes) - 1], IDX_ARRAY, np.arange(F, dtype=np.int64)), F, (K * K,))
    return LayerSpec("constant", F, 1, 0, K * K, {"num_output_units": K * K, "log_space": True}, {"value": value})


def squared_partition_plan(plan: Plan, *, conjugate: bool | None = None) -> Plan:
    """The plan of ``Z = integral of c(x) * conj(c(x)) dx`` for a circuit ``c`` made of Embedding
    inputs, Hadamard products and dense sums (as such or fused into CP-T layers).  Z has no
    variables: evaluate it with ``HipCircuit(z_plan, tensors_of_c)()`` -> ``(1, 1)``.  Categorical and Gaussian
    inputs are covered too (a real circuit squared: Z = sum / integral of c(x)^2).

    `conjugate` defaults to True under the complex semiring (|c|^2) and False otherwise (c^2)."""
    if conjugate is None:
        conjugate = plan.semiring == "complex-lse-sum"
    folds = [l.num_folds for l in plan.layers]
    out_layers: list[LayerSpec] = []
    z_folds: dict[int, int] = {}
    last_of: dict[int, int] = {}  # layer of c -> the layer of Z that carries its (squared) output

    def push(spec: LayerSpec) -> int:
        out_layers.append(spec)
        z_folds[len(out_layers) - 1] = spec.num_folds
        return len(out_layers) - 1

    def rows_of(l: LayerSpec) -> list[list[tuple[int, int]]]:
        pairs = resolve_fold_index(l.inputs, folds)  # (F, H, 2)
        return [[(last_of[int(p)], int(f)) for p, f in row] for row in pairs]

    for i, l in enumerate(plan.layers):
        F, Ki, Ko = l.num_folds, l.num_input_units, l.num_output_units
        if l.type == "embedding":
            if l.scope_idx is None or l.scope_idx.shape[1] != 1:
                raise NotImplementedError("integrating input layers over several variables")
            last_of[i] = push(_embedding_square_integral(l, conjugate))
            continue
        if l.type in ("categorical", "gaussian"):
            if l.scope_idx is None or l.scope_idx.shape[1] != 1:
                raise NotImplementedError("integrating input layers over several variables")
            last_of[i] = push(_categorical_square_integral(l) if l.type == "categorical" else _gaussian_square_integral(l))
            continue
        if l.type in ("hadamard", "cpt"):
            cur = push(LayerSpec("hadamard", F, l.arity, Ki * Ki, Ki * Ki, {"num_input_units": Ki * Ki, "arity": l.arity},
                                 {}, _fold_index(rows_of(l), z_folds)))
            if l.type == "hadamard":
                last_of[i] = cur
                continue
            src_rows = [[(cur, f)] for f in range(F)]
        elif l.type == "sum" and l.arity == 1:
            src_rows = rows_of(l)
        else:
            raise NotImplementedError(f"squaring {l.type!r} layers of arity {l.arity}")
        # (W x) (x) (W' x') = (W (x) W') (x (x) x'): two TensorDot layers, one per factor
        # (each contracts the slow factor of its input and emits it as the fast factor of its output)
        first = push(LayerSpec("tensordot", F, 1, Ki * Ki, Ki * Ko, {"num_input_units": Ki * Ki, "num_output_units": Ki * Ko},
                               {"weight": _pointer_graph(l.params["weight"], conjugate=False)}, _fold_index(src_rows, z_folds)))
        last_of[i] = push(LayerSpec("tensordot", F, 1, Ki * Ko, Ko * Ko, {"num_input_units": Ki * Ko, "num_output_units": Ko * Ko},
                                    {"weight": _pointer_graph(l.params["weight"], conjugate=conjugate)},
                                    _fold_index([[(first, f)] for f in range(F)], z_folds)))
    out_pairs = resolve_fold_index(plan.output, folds).reshape(-1, 2)
    output = _fold_index([[(last_of[int(p)], int(f)) for p, f in out_pairs]], z_folds, output=True)
    return Plan(plan.semiring, 0, out_layers, output, dict(plan.tensors), plan.name + "_Z")
