"""Host-side plan passes: layer fusion and folding of an UNFOLDED plan.

The reference lowers a symbolic circuit to one torch layer per symbolic layer, then (once per
circuit, host only) rewrites that graph:

* ``optimize``: fuse [Hadamard -> dense Sum] into a CP-T layer and [Kronecker -> dense Sum] into a
  Tucker layer (pattern definitions cirkit/backend/torch/optimization/layers.py:50-88, 201-279; the
  rebuilt graph keeps the topological order and puts the fused layer where the Sum was,
  graph/optimize.py:201-325);
* ``fold``: walk the layer-wise topological frontiers, stack the layers of a frontier that share
  type / configuration / parameter structure into one folded layer, and express every input as a
  stacked fold index (graph/folding.py:62-243, utils/algorithms.py:71-97).

`fuse_plan` and `fold_plan` are those two passes on `Plan`s (layers with one fold each in, folded
plan out), so a circuit extracted with ``fold=False, optimize=False`` -- or built by hand -- reaches
the same folded plan the reference would have produced; pinned by the reference's known-answer
circuits compiled under all four flag combinations (tests/test_compiler_passes.py).
"""

from __future__ import annotations

from typing import Any, Mapping

import numpy as np

from .plan import IDX_ARRAY, IDX_NONE, FoldIndex, LayerSpec, ParamGraph, ParamNode, Plan, resolve_fold_index
from .templates import _fold_index, _kahn_frontiers, _outgoings

_UNARY_PARAM_OPS = ("softmax", "log_softmax", "sigmoid", "scaled_sigmoid", "exp", "log", "square", "clamp", "softplus", "conj")


def _children(plan: Plan) -> list[list[tuple[int, int]] | None]:
    folds = [l.num_folds for l in plan.layers]
    out: list[list[tuple[int, int]] | None] = []
    for l in plan.layers:
        if l.inputs is None:
            out.append(None)
            continue
        if l.num_folds != 1:
            raise ValueError("this pass expects an unfolded plan (one fold per layer)")
        pairs = resolve_fold_index(l.inputs, folds).reshape(-1, 2)
        out.append([(int(p), int(f)) for p, f in pairs])
    return out


def _renumber_tensors(layers: list[LayerSpec], values: Mapping[str, Any], meta: Mapping[str, tuple]):
    """Name tensors t0, t1, ... in order of first use (what extracting the plan from a compiled
    circuit would give)."""
    rename: dict[str, str] = {}
    for l in layers:
        for pg in l.params.values():
            for n in pg.nodes:
                if n.op in ("tensor", "pointer"):
                    old = n.config["tensor"]
                    if old not in rename:
                        rename[old] = f"t{len(rename)}"
    for l in layers:
        for pg in l.params.values():
            for n in pg.nodes:
                if n.op in ("tensor", "pointer"):
                    n.config = {**n.config, "tensor": rename[n.config["tensor"]]}
    return ({rename[k]: values[k] for k in rename}, {rename[k]: meta[k] for k in rename})


def _copy_layer(l: LayerSpec) -> LayerSpec:
    params = {
        pn: ParamGraph(
            [ParamNode(n.op, n.num_folds, tuple(n.shape), dict(n.config), list(n.inputs)) for n in pg.nodes],
            pg.output, pg.num_folds, tuple(pg.shape),
        )
        for pn, pg in l.params.items()
    }
    return LayerSpec(l.type, l.num_folds, l.arity, l.num_input_units, l.num_output_units, dict(l.config), params,
                     l.inputs, None if l.scope_idx is None else np.array(l.scope_idx))


# ---------------------------------------------------------------------------------------------
# fusion
# ---------------------------------------------------------------------------------------------
def fuse_plan(plan: Plan, tensors: Mapping[str, Any]) -> tuple[Plan, dict[str, Any]]:
    """[Hadamard -> Sum(arity 1)] => CP-T and [Kronecker -> Sum(arity 1)] => Tucker on an unfolded plan."""
    ch = _children(plan)
    n = len(plan.layers)
    ins = {j: [p for p, _ in ch[j]] for j in range(n) if ch[j] is not None}
    outs = _outgoings(list(range(n)), ins)
    out_ids = {int(p) for p, _ in resolve_fold_index(plan.output, [l.num_folds for l in plan.layers]).reshape(-1, 2)}
    fused_into: dict[int, int] = {}  # product layer -> sum layer that absorbs it
    for j, l in enumerate(plan.layers):
        if l.type == "sum" and l.arity == 1 and ch[j] is not None:
            (p, _), = ch[j]
            src = plan.layers[p]
            if src.type in ("hadamard", "kronecker") and len(outs.get(p, ())) == 1 and p not in out_ids:
                fused_into[p] = j
    absorbed = {j: p for p, j in fused_into.items()}
    new_id: dict[int, int] = {}
    layers: list[LayerSpec] = []
    for j, l in enumerate(plan.layers):
        if j in fused_into:
            continue
        nl = _copy_layer(l)
        src_children = ch[j]
        if j in absorbed:
            prod = plan.layers[absorbed[j]]
            src_children = ch[absorbed[j]]
            kind = "cpt" if prod.type == "hadamard" else "tucker"
            nl = LayerSpec(kind, 1, prod.arity, prod.num_input_units, l.num_output_units,
                           {"num_input_units": prod.num_input_units, "num_output_units": l.num_output_units,
                            "arity": prod.arity}, nl.params, None, None)
        if src_children is not None:
            ids = list(dict.fromkeys(new_id[p] for p, _ in src_children))
            rows = [[(new_id[p], f) for p, f in src_children]]
            nl.inputs = _fold_index(rows, {i: 1 for i in ids})
        new_id[j] = len(layers)
        layers.append(nl)
    out_pairs = resolve_fold_index(plan.output, [l.num_folds for l in plan.layers]).reshape(-1, 2)
    output = _fold_index([[(new_id[int(p)], int(f)) for p, f in out_pairs]], {i: 1 for i in range(len(layers))}, output=True)
    vals, meta = _renumber_tensors(layers, tensors, plan.tensors)
    return Plan(plan.semiring, plan.num_variables, layers, output, meta, plan.name), vals


# ---------------------------------------------------------------------------------------------
# folding
# ---------------------------------------------------------------------------------------------
def _param_signature(pg: ParamGraph):
    sig = []
    for n in pg.nodes:
        cfg = tuple(sorted((k, str(v)) for k, v in n.config.items() if k not in ("tensor",)))
        sig.append((n.op, tuple(n.shape), cfg, tuple(tuple(fi.ids) for fi in n.inputs)))
    return tuple(sig)


def _fold_params(group: list[LayerSpec], tensors: Mapping[str, Any], new_vals: dict, new_meta: dict):
    F = len(group)
    out: dict[str, ParamGraph] = {}
    for pn, pg0 in group[0].params.items():
        nodes: list[ParamNode] = []
        for k, n0 in enumerate(pg0.nodes):
            if n0.op == "tensor":
                vals = []
                for l in group:
                    nk = l.params[pn].nodes[k]
                    v = np.asarray(tensors[nk.config["tensor"]])
                    if v.shape[0] != 1:
                        raise ValueError("fold_plan expects one fold per tensor")
                    vals.append(v)
                name = f"f{len(new_vals)}"
                new_vals[name] = np.concatenate(vals, axis=0)
                new_meta[name] = (tuple(new_vals[name].shape), str(new_vals[name].dtype))
                nodes.append(ParamNode("tensor", F, tuple(n0.shape), {"tensor": name}, []))
            elif n0.op in _UNARY_PARAM_OPS:
                nodes.append(ParamNode(n0.op, F, tuple(n0.shape), dict(n0.config), [FoldIndex(list(n0.inputs[0].ids), IDX_NONE)]))
            else:
                raise NotImplementedError(f"folding parameter graphs with {n0.op!r} nodes")
        out[pn] = ParamGraph(nodes, FoldIndex(list(pg0.output.ids), IDX_ARRAY, np.arange(F, dtype=np.int64)), F, tuple(pg0.shape))
    return out


def fold_plan(plan: Plan, tensors: Mapping[str, Any]) -> tuple[Plan, dict[str, Any]]:
    """Stack the layers of each layer-wise frontier that share type, configuration and parameter
    structure; returns the folded plan and the stacked parameter values."""
    ch = _children(plan)
    n = len(plan.layers)
    ins = {j: [p for p, _ in ch[j]] for j in range(n) if ch[j] is not None}
    fronts = _kahn_frontiers(list(range(n)), ins)
    layers: list[LayerSpec] = []
    folds: dict[int, int] = {}
    where: dict[int, tuple[int, int]] = {}
    new_vals: dict[str, Any] = {}
    new_meta: dict[str, tuple] = {}
    for frontier in fronts:
        groups: dict[tuple, list[int]] = {}
        for j in frontier:
            l = plan.layers[j]
            key = (l.type, l.arity, l.num_input_units, l.num_output_units,
                   tuple((k, str(v)) for k, v in l.config.items()),
                   None if l.scope_idx is None else l.scope_idx.shape[1],
                   tuple((pn, _param_signature(pg)) for pn, pg in l.params.items()))
            groups.setdefault(key, []).append(j)
        for members in groups.values():
            mid = len(layers)
            for i, j in enumerate(members):
                where[j] = (mid, i)
            folds[mid] = len(members)
            first = plan.layers[members[0]]
            group = [plan.layers[j] for j in members]
            params = _fold_params(group, tensors, new_vals, new_meta)
            scope = None if first.scope_idx is None else np.concatenate([l.scope_idx for l in group], axis=0)
            inputs = None
            if ch[members[0]] is not None:
                rows = [[where[p] for p, _ in ch[j]] for j in members]
                inputs = _fold_index(rows, folds)
            layers.append(LayerSpec(first.type, len(members), first.arity, first.num_input_units,
                                    first.num_output_units, dict(first.config), params, inputs, scope))
    out_pairs = resolve_fold_index(plan.output, [l.num_folds for l in plan.layers]).reshape(-1, 2)
    output = _fold_index([[where[int(p)] for p, _ in out_pairs]], folds, output=True)
    vals, meta = _renumber_tensors(layers, new_vals, new_meta)
    return Plan(plan.semiring, plan.num_variables, layers, output, meta, plan.name), vals


def compile_plan(plan: Plan, tensors: Mapping[str, Any], *, fold: bool = True, optimize: bool = True):
    """The reference's post-processing order (torch/compiler.py:311-332): optimise, then fold."""
    if optimize:
        plan, tensors = fuse_plan(plan, tensors)
    if fold:
        plan, tensors = fold_plan(plan, tensors)
    return plan, dict(tensors)
