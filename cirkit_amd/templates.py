"""Native plan builders: region graph -> folded, optimised evaluation plan, without cirkit.

SURVEY.md section 8 (f1).  The reference gets from a region-graph template to a folded
``TorchCircuit`` through five host-side stages; each is re-derived here on plain integers so that the
result is THE SAME plan the reference produces (same folded layer order, fold order, index arrays
and parameter-tensor order), which `tests/test_templates.py` pins against the committed plan
fixtures extracted from the real reference:

1. region graph            QuadTree  cirkit/templates/region_graph/algorithms/quad.py:62-193
                           RandomBinaryTree  .../algorithms/random.py:17-109
2. layers per region       RegionGraph.build_circuit ('cp' / 'cp-t')  region_graph/graph.py:344-588
3. compile order           Kahn ordering of the layer DAG  utils/algorithms.py:47-68,
                           torch/compiler.py:257-309
4. fusion                  Hadamard -> dense Sum  =>  CP-T layer (CandecompPattern,
                           optimization/layers.py:70-88, 260-279; graph rebuild optimize.py:201-325)
5. folding                 layer-wise frontiers, grouping by fold settings, stacked fold indices
                           (utils/algorithms.py:71-97, graph/folding.py:62-243)

Only what the image/tabular templates of the BASELINE configurations use is covered: one variable
per input region, binary partitions, one partition per region (tree-shaped region graphs).
"""

from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from .plan import IDX_ARRAY, IDX_NONE, IDX_UNSQ0, IDX_UNSQ1, FoldIndex, LayerSpec, ParamGraph, ParamNode, Plan


# ---------------------------------------------------------------------------------------------
# graph orderings (Kahn's algorithm, FIFO, consumers in node-list order)
# ---------------------------------------------------------------------------------------------
def _outgoings(nodes, ins):
    out: dict[Any, list] = {}
    for n in nodes:
        for ch in ins.get(n, ()):
            out.setdefault(ch, []).append(n)
    return out


def _kahn(nodes, ins):
    out = _outgoings(nodes, ins)
    pending = {n: len(ins.get(n, ())) for n in nodes}
    queue = deque(n for n in nodes if pending[n] == 0)
    order = []
    while queue:
        n = queue.popleft()
        order.append(n)
        for m in out.get(n, ()):
            pending[m] -= 1
            if pending[m] == 0:
                queue.append(m)
    if any(pending.values()):
        raise ValueError("cycle in graph")
    return order


def _kahn_frontiers(nodes, ins):
    out = _outgoings(nodes, ins)
    pending = {n: len(ins.get(n, ())) for n in nodes}
    frontier = [n for n in nodes if pending[n] == 0]
    fronts = [frontier]
    while True:
        nxt = []
        for n in frontier:
            for m in out.get(n, ()):
                pending[m] -= 1
                if pending[m] == 0:
                    nxt.append(m)
        if not nxt:
            break
        fronts.append(nxt)
        frontier = nxt
    if any(pending.values()):
        raise ValueError("cycle in graph")
    return fronts


# ---------------------------------------------------------------------------------------------
# region graphs
# ---------------------------------------------------------------------------------------------
@dataclass
class RegionGraph:
    """Nodes are ints; `scope[n]` the variables of node n; regions and partitions alternate."""

    nodes: list[int]
    ins: dict[int, list[int]]
    is_region: dict[int, bool]
    scope: dict[int, tuple[int, ...]]
    root: int


def quad_tree(height: int, width: int, *, num_patch_splits: int = 2) -> RegionGraph:
    """Quad-tree region graph over a (1, H, W) image: pixels are merged frontier by frontier, 2x2
    blocks first horizontally then vertically (num_patch_splits = 2) or at once (4); odd borders
    pass single regions up unchanged."""
    if height <= 0 or width <= 0:
        raise ValueError("The height and the width must be positive")
    if num_patch_splits not in (2, 4):
        raise ValueError("The number of patches to split must be either 2 or 4")
    nodes: list[int] = []
    ins: dict[int, list[int]] = {}
    is_region: dict[int, bool] = {}
    scope: dict[int, tuple[int, ...]] = {}

    def new(region: bool, sc) -> int:
        n = len(is_region)
        is_region[n] = region
        scope[n] = tuple(sorted(sc))
        nodes.append(n)
        return n

    grid = [[new(True, (i * width + j,)) for j in range(width)] for i in range(height)]

    def merge(parts: list[int]) -> int:
        sc = [v for p in parts for v in scope[p]]
        rgn = new(True, sc)
        ptn = new(False, sc)
        ins[rgn] = [ptn]
        ins[ptn] = list(parts)
        return rgn

    h, w = height, width
    while h > 1 or w > 1:
        nh, nw = (h + 1) // 2, (w + 1) // 2
        new_grid = [[-1] * nw for _ in range(nh)]
        for i in range(nh):
            for j in range(nw):
                cells = [
                    grid[a][b]
                    for a, b in ((2 * i, 2 * j), (2 * i, 2 * j + 1), (2 * i + 1, 2 * j), (2 * i + 1, 2 * j + 1))
                    if a < h and b < w
                ]
                if len(cells) == 1:
                    node = cells[0]
                elif len(cells) == 2:
                    node = merge(cells)
                elif num_patch_splits == 2:
                    node = merge([merge(cells[:2]), merge(cells[2:])])
                else:
                    node = merge(cells)
                new_grid[i][j] = node
        grid, h, w = new_grid, nh, nw
    return RegionGraph(nodes, ins, is_region, scope, grid[0][0])


def random_binary_tree(num_variables: int, *, depth: int | None = None, seed: int = 42) -> RegionGraph:
    """Random balanced binary partitioning of {0..n-1} (one repetition), drawn with
    ``numpy.random.RandomState(seed).shuffle`` exactly like the reference template."""
    if num_variables <= 0:
        raise ValueError("The number of variables must be positive")
    max_depth = int(np.ceil(np.log2(num_variables)))
    if depth is None:
        depth = max_depth
    elif depth < 0 or depth > max_depth:
        raise ValueError(f"The depth must be between 0 and {max_depth}")
    rs = np.random.RandomState(seed)
    nodes: list[int] = []
    ins: dict[int, list[int]] = {}
    is_region: dict[int, bool] = {}
    scope: dict[int, tuple[int, ...]] = {}
    # The reference keeps a scope as a frozenset and shuffles `list(scope)`, i.e. the list starts in
    # CPython's set-iteration order (NOT sorted for small sets such as {9, 2, 5}).  To draw the same
    # tree the same container is used here; the result is still a pure function of (n, depth, seed).
    as_set: dict[int, frozenset] = {}

    def new(region: bool, sc) -> int:
        n = len(is_region)
        is_region[n] = region
        sc = list(sc)
        scope[n] = tuple(sc)
        as_set[n] = frozenset(sc)
        nodes.append(n)
        return n

    root = new(True, range(num_variables))
    frontier = [root]
    for _ in range(depth):
        nxt: list[int] = []
        for rgn in frontier:
            ls = list(as_set[rgn])
            rs.shuffle(ls)
            cut = int(np.round(0.5 * len(ls)))
            parts = [p for p in (ls[:cut], ls[cut:]) if p]
            if len(parts) == 1:
                continue
            ptn = new(False, scope[rgn])
            kids = [new(True, p) for p in parts]
            ins.setdefault(rgn, []).append(ptn)
            ins[ptn] = kids
            nxt.extend(kids)
        frontier = nxt
    return RegionGraph(nodes, ins, is_region, scope, root)


# ---------------------------------------------------------------------------------------------
# symbolic layers -> compile order -> fusion -> folding
# ---------------------------------------------------------------------------------------------
@dataclass(eq=False)
class _L:
    kind: str  # 'input' | 'sum' | 'hadamard' | 'cpt'
    ki: int
    ko: int
    arity: int = 1
    var: int = -1  # input layers
    pid: int = -1  # parameter identity (unfolded), -1 if none


@dataclass
class InputSpec:
    """Input-layer family: ``categorical`` (probs = softmax over categories), ``embedding``
    (weight, no activation) or ``gaussian`` (mean, stddev = scaled sigmoid)."""

    name: str = "categorical"
    num_states: int = 256


def _layers_of_region_graph(rg: RegionGraph, inp: InputSpec, sum_product: str, k_in: int, k_sum: int, num_classes: int):
    if sum_product not in ("cp", "cp-t"):
        raise NotImplementedError(f"sum-product layer {sum_product!r} (supported: 'cp', 'cp-t')")
    layers: list[_L] = []
    ins: dict[_L, list[_L]] = {}
    of: dict[int, _L] = {}
    outs = _outgoings(rg.nodes, rg.ins)
    npid = [0]

    def pid() -> int:
        npid[0] += 1
        return npid[0] - 1

    for node in _kahn(rg.nodes, rg.ins):
        if not rg.is_region[node]:
            continue
        parts = rg.ins.get(node, [])
        is_root = not outs.get(node)
        if not parts:
            if len(rg.scope[node]) != 1:
                raise NotImplementedError("input regions over more than one variable")
            l = _L("input", 1, k_in, var=rg.scope[node][0], pid=pid())
            layers.append(l)
            of[node] = l
            continue
        if len(parts) != 1:
            raise NotImplementedError("regions with several partitionings (mixing layers)")
        kids = [of[r] for r in rg.ins[parts[0]]]
        if sum_product == "cp":
            denses = [_L("sum", c.ko, k_sum, pid=pid()) for c in kids]
            had = _L("hadamard", k_sum, k_sum, arity=len(kids))
            layers.extend(denses)
            layers.append(had)
            ins[had] = denses
            for d, c in zip(denses, kids):
                ins[d] = [c]
            if not is_root:
                of[node] = had
            else:
                od = _L("sum", k_sum, num_classes, pid=pid())
                layers.append(od)
                ins[od] = [had]
                of[node] = od
        else:
            units = {c.ko for c in kids}
            if len(units) > 1:
                raise ValueError("Cannot build a CP transposed layer, as the inputs would have different units")
            ku = units.pop()
            had = _L("hadamard", ku, ku, arity=len(kids))
            dense = _L("sum", ku, num_classes if is_root else k_sum, pid=pid())
            layers.extend([had, dense])
            ins[had] = kids
            ins[dense] = [had]
            of[node] = dense
    return layers, ins, of[rg.root]


def _fuse_candecomp(order: list[_L], ins: dict[_L, list[_L]], output: _L):
    """[Hadamard -> dense Sum (arity 1)] => CP-T, inserted at the Sum's position."""
    outs = _outgoings(order, ins)
    had_of_sum: dict[_L, _L] = {}
    for l in order:
        if l.kind == "sum" and l.arity == 1:
            (src,) = ins[l]
            if src.kind == "hadamard" and len(outs.get(src, ())) == 1:
                had_of_sum[l] = src
    matched_h = set(had_of_sum.values())
    new_of: dict[_L, _L] = {}
    modules: list[_L] = []
    new_ins: dict[_L, list[_L]] = {}
    for l in order:
        if l in matched_h:
            continue
        if l in had_of_sum:
            h = had_of_sum[l]
            cpt = _L("cpt", h.ki, l.ko, arity=h.arity, pid=l.pid)
            new_of[l] = cpt
            modules.append(cpt)
            new_ins[cpt] = [new_of.get(c, c) for c in ins.get(h, ())]
        else:
            modules.append(l)
            new_ins[l] = [new_of.get(c, c) for c in ins.get(l, ())]
    return modules, new_ins, new_of.get(output, output)


def _fold_index(rows: list[list[tuple[int, int]]], folds: dict[int, int], *, output: bool = False) -> FoldIndex:
    """Stacked address-book entry (graph/folding.py:202-243): concatenate the distinct producers in
    order of first use; an identity pattern becomes an unsqueeze shortcut."""
    ids = list(dict.fromkeys(p for row in rows for p, _ in row))
    sizes = [folds[i] for i in ids]
    base = dict(zip(ids, np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(int)))
    cum = [[int(base[p]) + f for p, f in row] for row in rows]
    if output:
        return FoldIndex(ids, IDX_ARRAY, np.asarray(cum[0], dtype=np.int64))
    total = int(sum(sizes))
    if [i for row in cum for i in row] == list(range(total)):
        if len(cum) == 1 and len(cum[0]) == total:
            return FoldIndex(ids, IDX_UNSQ0)
        if len(cum) == total and len(cum[0]) == 1:
            return FoldIndex(ids, IDX_UNSQ1)
    return FoldIndex(ids, IDX_ARRAY, np.asarray(cum, dtype=np.int64))


def _param_graph(F: int, shape: tuple[int, ...], tensor: str, activation: str) -> ParamGraph:
    nodes = [ParamNode("tensor", F, shape, {"tensor": tensor}, [])]
    if activation == "softmax":
        nodes.append(ParamNode("softmax", F, shape, {"dim": len(shape) - 1}, [FoldIndex([0], IDX_NONE)]))
    elif activation == "scaled-sigmoid":
        nodes.append(ParamNode("scaled_sigmoid", F, shape, {"vmin": 1e-05, "vmax": 1.0}, [FoldIndex([0], IDX_NONE)]))
    elif activation != "none":
        raise NotImplementedError(f"activation {activation!r}")
    return ParamGraph(nodes, FoldIndex([len(nodes) - 1], IDX_ARRAY, np.arange(F, dtype=np.int64)), F, shape)


def build_plan(
    rg: RegionGraph,
    *,
    input_layer: InputSpec | str = "categorical",
    sum_product: str = "cp",
    num_input_units: int = 32,
    num_sum_units: int = 32,
    num_classes: int = 1,
    sum_activation: str = "softmax",
    input_activation: str | None = None,
    semiring: str = "lse-sum",
    name: str = "",
) -> Plan:
    """Region graph -> the folded, optimised plan ``compile(fold=True, optimize=True)`` would give."""
    inp = InputSpec(input_layer) if isinstance(input_layer, str) else input_layer
    if inp.name not in ("categorical", "embedding", "gaussian"):
        raise NotImplementedError(f"input layer {inp.name!r}")
    layers, ins, out = _layers_of_region_graph(rg, inp, sum_product, num_input_units, num_sum_units, num_classes)
    order = _kahn(layers, ins)  # compile order (torch/compiler.py:275-282)
    modules, mins, out = _fuse_candecomp(order, ins, out)
    # a second optimisation round finds nothing new for these templates
    fronts = _kahn_frontiers(modules, mins)

    specs: list[LayerSpec] = []
    folds: dict[int, int] = {}
    where: dict[_L, tuple[int, int]] = {}
    tensors: dict[str, tuple[tuple[int, ...], str]] = {}

    def new_tensor(shape) -> str:
        t = f"t{len(tensors)}"
        tensors[t] = (tuple(int(s) for s in shape), "float32")
        return t

    for frontier in fronts:
        groups: dict[tuple, list[_L]] = {}
        for l in frontier:
            groups.setdefault((l.kind, l.ki, l.ko, l.arity), []).append(l)
        for (kind, ki, ko, arity), group in groups.items():
            F = len(group)
            mid = len(specs)
            for i, l in enumerate(group):
                where[l] = (mid, i)
            folds[mid] = F
            if kind == "input":
                scope = np.asarray([[l.var] for l in group], dtype=np.int64)
                if inp.name == "categorical":
                    act = input_activation or "softmax"
                    cfg = {"num_output_units": ko, "num_categories": inp.num_states}
                    pname = "probs" if act == "softmax" else "logits"
                    params = {pname: _param_graph(F, (ko, inp.num_states), new_tensor((F, ko, inp.num_states)), act)}
                    spec = LayerSpec("categorical", F, 1, 1, ko, cfg, params, None, scope)
                elif inp.name == "embedding":
                    cfg = {"num_output_units": ko, "num_states": inp.num_states}
                    params = {"weight": _param_graph(F, (ko, inp.num_states), new_tensor((F, ko, inp.num_states)),
                                                     input_activation or "none")}
                    spec = LayerSpec("embedding", F, 1, 1, ko, cfg, params, None, scope)
                else:
                    params = {
                        "mean": _param_graph(F, (ko,), new_tensor((F, ko)), "none"),
                        "stddev": _param_graph(F, (ko,), new_tensor((F, ko)), "scaled-sigmoid"),
                    }
                    spec = LayerSpec("gaussian", F, 1, 1, ko, {"num_output_units": ko}, params, None, scope)
                specs.append(spec)
                continue
            rows = [[where[c] for c in mins[l]] for l in group]
            if kind in ("sum", "cpt"):
                rows_idx = rows if kind == "cpt" else rows  # sum (arity 1): one child per fold
                fi = _fold_index(rows_idx, folds)
                cfg = {"num_input_units": ki, "num_output_units": ko, "arity": arity}
                wshape = (ko, ki * (arity if kind == "sum" else 1))
                params = {"weight": _param_graph(F, wshape, new_tensor((F, *wshape)), sum_activation)}
                specs.append(LayerSpec(kind, F, arity, ki, ko, cfg, params, fi))
            else:  # hadamard left unfused
                fi = _fold_index(rows, folds)
                specs.append(LayerSpec("hadamard", F, arity, ki, ko, {"num_input_units": ki, "arity": arity}, {}, fi))
    output = _fold_index([[where[out]]], folds, output=True)
    nvars = max(max(s) for s in rg.scope.values()) + 1
    return Plan(semiring, nvars, specs, output, tensors, name)


def quad_tree_plan(shape: tuple[int, int, int] = (1, 28, 28), **kw: Any) -> Plan:
    """``image_data(shape, 'quad-tree-2', ...)`` + ``compile(fold=True, optimize=True)``."""
    c, h, w = shape
    if c != 1:
        raise NotImplementedError("multi-channel images (factorised multivariate inputs)")
    return build_plan(quad_tree(h, w, num_patch_splits=kw.pop("num_patch_splits", 2)), **kw)


def random_binary_tree_plan(num_features: int, *, depth: int | None = None, seed: int = 42, **kw: Any) -> Plan:
    """``tabular_data('random-binary-tree', num_features=...)`` + compile."""
    return build_plan(random_binary_tree(num_features, depth=depth, seed=seed), **kw)


# ---------------------------------------------------------------------------------------------
# the reference's template entry points (cirkit/templates/data_modalities.py:26-305), same argument
# names, returning the compiled plan instead of a symbolic circuit
# ---------------------------------------------------------------------------------------------
def image_data(
    image_shape: tuple[int, int, int],
    *,
    region_graph: str = "quad-tree-2",
    input_layer: str = "categorical",
    num_input_units: int,
    sum_product_layer: str = "cp",
    num_sum_units: int,
    num_classes: int = 1,
    sum_weight_activation: str = "softmax",
    input_activation: str | None = None,
    semiring: str = "lse-sum",
) -> Plan:
    """``data_modalities.image_data`` (:26-162) followed by ``compile(fold=True, optimize=True)``."""
    if region_graph not in ("quad-tree-2", "quad-tree-4"):
        raise NotImplementedError(f"region graph {region_graph!r} (native builders: 'quad-tree-2', 'quad-tree-4')")
    states = 256  # data_modalities.py:120-126: categorical / embedding over 8-bit pixels
    return quad_tree_plan(
        image_shape, num_patch_splits=int(region_graph[-1]), input_layer=InputSpec(input_layer, states),
        sum_product=sum_product_layer, num_input_units=num_input_units, num_sum_units=num_sum_units,
        num_classes=num_classes, sum_activation=sum_weight_activation, input_activation=input_activation,
        semiring=semiring,
    )


def tabular_data(
    region_graph: str = "random-binary-tree",
    *,
    num_features: int,
    input_layers: dict,
    num_input_units: int,
    sum_product_layer: str = "cp",
    num_sum_units: int,
    num_classes: int = 1,
    semiring: str = "lse-sum",
) -> Plan:
    """``data_modalities.tabular_data`` (:165-305) followed by ``compile(fold=True, optimize=True)``."""
    if region_graph != "random-binary-tree":
        raise NotImplementedError(f"region graph {region_graph!r} (native builder: 'random-binary-tree')")
    args = dict(input_layers.get("args", {}))
    name = input_layers["name"]
    states = int(args.get("num_categories", args.get("num_states", 2)))
    return random_binary_tree_plan(
        num_features, input_layer=InputSpec(name, states), sum_product=sum_product_layer,
        num_input_units=num_input_units, num_sum_units=num_sum_units, num_classes=num_classes, semiring=semiring,
    )
