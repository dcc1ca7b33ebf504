"""Native plan builders: region graph -> folded, optimised evaluation plan, without cirkit.

SURVEY.md section 8 (f1).  The reference gets from a region-graph template to a folded
``TorchCircuit`` through five host-side stages; each is re-derived here on plain integers so that the
result is THE SAME plan the reference produces (same folded layer order, fold order, index arrays
and parameter-tensor order), which `tests/test_templates.py` pins against the committed plan
fixtures extracted from the real reference:

1. region graph            QuadTree / QuadGraph  cirkit/templates/region_graph/algorithms/quad.py:62-193
                           RandomBinaryTree  .../algorithms/random.py:17-109
                           PoonDomingos  .../algorithms/poon_domingos.py:18-205
2. layers per region       RegionGraph.build_circuit ('cp' / 'cp-t' / 'tucker', mixing layers for regions
                           with several partitionings, factorised multivariate inputs)
                           region_graph/graph.py:344-588, symbolic/parameters.py:1007-1044
3. compile order           Kahn ordering of the layer DAG  utils/algorithms.py:47-68,
                           torch/compiler.py:257-309
4. fusion                  up to five rounds of pattern matching over the layer DAG: Sum -> Sum
                           (SumCollapse, weights multiplied), Kronecker -> Sum (Tucker), Hadamard -> Sum
                           (CP-T); pattern order, conflict resolution and graph rebuild as in
                           optimization/layers.py:30-88,162-279,468-472, graph/optimize.py:201-470,
                           torch/compiler.py:509-556,668-772
5. folding                 layer-wise frontiers, grouping by fold settings, stacked fold indices, and
                           the same procedure one level down for the parameter graphs of a group
                           (utils/algorithms.py:71-97, graph/folding.py:62-298, torch/compiler.py:335-506)

Binomial input layers are built here as well (round 1; outside SURVEY.md section 8's f1 list -- kept because committed
plan fixtures use them, not extended).  The Chow-Liu structure learner of round 1 is gone (out of scope).
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
    """Nodes are ints; `scope[n]` the variables of node n; regions and partitions alternate.
    `leaf_vars[n]` lists a region's variables in the order the reference iterates its scope (see
    `_scope_order`), which fixes the order of the factorised input layers of a multivariate input
    region."""

    nodes: list[int]
    ins: dict[int, list[int]]
    is_region: dict[int, bool]
    scope: dict[int, tuple[int, ...]]
    root: int
    leaf_vars: dict[int, list[int]] = field(default_factory=dict)

    def variables_of_leaf(self, n: int) -> list[int]:
        return list(self.leaf_vars.get(n, self.scope[n]))


def _scope_order(variables, wraps: int = 2) -> list[int]:
    """The reference keeps a scope as a frozenset and iterates THAT, so the order of a region's
    variables is CPython's set-iteration order (not sorted once hashes collide modulo the table
    size).  A region node re-wraps the scope it is given (Scope(Scope(list))), hence two rounds by
    default; the result is a pure function of the variable list."""
    vs = list(variables)
    for _ in range(wraps):
        vs = list(frozenset(vs))
    return vs


class _RGBuilder:
    def __init__(self) -> None:
        self.nodes: list[int] = []
        self.ins: dict[int, list[int]] = {}
        self.is_region: dict[int, bool] = {}
        self.scope: dict[int, tuple[int, ...]] = {}
        self.leaf_vars: dict[int, list[int]] = {}

    def new(self, region: bool, sc, *, sort: bool = True) -> int:
        n = len(self.is_region)
        self.is_region[n] = region
        self.scope[n] = tuple(sorted(sc)) if sort else tuple(sc)
        self.nodes.append(n)
        return n

    def graph(self, root: int) -> RegionGraph:
        return RegionGraph(self.nodes, self.ins, self.is_region, self.scope, root, self.leaf_vars)


def _quad(height: int, width: int, num_channels: int, *, is_tree: bool, num_patch_splits: int) -> RegionGraph:
    if num_channels <= 0 or height <= 0 or width <= 0:
        raise ValueError("The number of channels, the height and the width must be positive")
    if is_tree and num_patch_splits not in (2, 4):
        raise ValueError("The number of patches to split must be either 2 or 4")
    b = _RGBuilder()
    hw = height * width

    def pixel(i: int, j: int) -> int:
        vs = [c * hw + i * width + j for c in range(num_channels)]
        n = b.new(True, vs)
        b.leaf_vars[n] = _scope_order(vs)
        return n

    grid = [[pixel(i, j) for j in range(width)] for i in range(height)]

    def merge(parts: list[int]) -> int:
        sc = [v for p in parts for v in b.scope[p]]
        rgn = b.new(True, sc)
        ptn = b.new(False, sc)
        b.ins[rgn] = [ptn]
        b.ins[ptn] = list(parts)
        return rgn

    def merge4_dag(c: list[int]) -> int:
        # the whole block split top/bottom AND left/right: two partitionings of the same region
        rgn = merge([merge(c[:2]), merge(c[2:])])
        left, right = merge([c[0], c[2]]), merge([c[1], c[3]])
        ptn = b.new(False, b.scope[rgn])
        b.ins[ptn] = [left, right]
        b.ins[rgn].append(ptn)
        return rgn

    h, w = height, width
    while h > 1 or w > 1:
        nh, nw = (h + 1) // 2, (w + 1) // 2
        new_grid = [[-1] * nw for _ in range(nh)]
        for i in range(nh):
            for j in range(nw):
                cells = [
                    grid[a][c]
                    for a, c in ((2 * i, 2 * j), (2 * i, 2 * j + 1), (2 * i + 1, 2 * j), (2 * i + 1, 2 * j + 1))
                    if a < h and c < w
                ]
                if len(cells) == 1:
                    node = cells[0]
                elif len(cells) == 2:
                    node = merge(cells)
                elif not is_tree:
                    node = merge4_dag(cells)
                elif num_patch_splits == 2:
                    node = merge([merge(cells[:2]), merge(cells[2:])])
                else:
                    node = merge(cells)
                new_grid[i][j] = node
        grid, h, w = new_grid, nh, nw
    return b.graph(grid[0][0])


def quad_tree(height: int, width: int, *, num_patch_splits: int = 2, num_channels: int = 1) -> RegionGraph:
    """Quad-tree region graph over a (C, H, W) image: pixels are merged frontier by frontier, 2x2
    blocks first horizontally then vertically (num_patch_splits = 2) or at once (4); odd borders
    pass single regions up unchanged.  A pixel region holds its C channel variables."""
    return _quad(height, width, num_channels, is_tree=True, num_patch_splits=num_patch_splits)


def quad_graph(height: int, width: int, *, num_channels: int = 1) -> RegionGraph:
    """Quad-graph: like the quad-tree with two splits, but every full 2x2 block is decomposed both
    top/bottom and left/right, so its region has two partitionings (a mixing layer in the circuit)."""
    return _quad(height, width, num_channels, is_tree=False, num_patch_splits=2)


def _pd_cut_points(delta, shape: tuple[int, int, int]) -> list[list[list[int]]]:
    if isinstance(delta, (float, int)):
        delta = [delta]
    levels = [[d, d] if isinstance(d, (float, int)) else list(d) for d in delta]
    if any(len(d) != 2 for d in levels):
        raise ValueError("Each delta list must be of same length as axes.")
    if any(dx < 1 for d in levels for dx in d):
        raise ValueError("Each delta must be >=1.")
    return [[[int((j + 1) * dx) for j in range(int((shape[ax] - 1) // dx))] for ax, dx in zip((1, 2), d)] for d in levels]


def poon_domingos(shape: tuple[int, int, int], *, delta, max_depth: int | None = None) -> RegionGraph:
    """Poon-Domingos region graph: starting from the whole image, every rectangle is cut at every
    admissible multiple of `delta` along the height and the width (one partitioning per cut);
    rectangles are shared between cuts, explored breadth first, coarsest delta level first."""
    num_channels, height, width = shape
    levels = _pd_cut_points(delta, shape)
    if max_depth is None:
        max_depth = sum(shape) + 1
    b = _RGBuilder()
    hw = height * width
    region_of: dict[tuple[int, int, int, int], int] = {}

    def region(cube: tuple[int, int, int, int]) -> int:
        if cube not in region_of:
            y0, x0, y1, x1 = cube
            vs = [c * hw + i * width + j for c in range(num_channels) for i in range(y0, y1) for j in range(x0, x1)]
            n = b.new(True, vs)
            b.leaf_vars[n] = _scope_order(vs)
            region_of[cube] = n
        return region_of[cube]

    root_cube = (0, 0, height, width)
    root = region(root_cube)
    depth = {root_cube: 0}
    queue = deque([root_cube])
    while queue:
        cube = queue.popleft()
        if depth[cube] > max_depth:
            continue
        found = False
        for per_axis in levels:
            for ax, cuts in enumerate(per_axis):  # 0: height, 1: width
                for cut in cuts:
                    if not cube[ax] < cut < cube[ax + 2]:
                        continue
                    found = True
                    lo, hi = list(cube), list(cube)
                    lo[ax + 2] = cut
                    hi[ax] = cut
                    halves = [tuple(lo), tuple(hi)]
                    rgn = region(cube)
                    kids = [region(c) for c in halves]
                    ptn = b.new(False, b.scope[rgn])
                    b.ins.setdefault(rgn, []).append(ptn)
                    b.ins[ptn] = kids
                    for c in halves:
                        if c not in depth:
                            depth[c] = depth[cube] + 1
                            queue.append(c)
            if found:
                break
    return b.graph(root)


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
    b = _RGBuilder()
    # The reference shuffles `list(scope)`, i.e. the list starts in set-iteration order (NOT sorted
    # for small sets such as {9, 2, 5}); the same order is used here, so the same tree is drawn and
    # the result is still a pure function of (n, depth, seed).

    def new(region: bool, sc, wraps: int = 2) -> int:
        sc = _scope_order(sc, wraps)
        n = b.new(region, sc, sort=False)
        b.leaf_vars[n] = sc
        return n

    root = new(True, range(num_variables), wraps=1)
    frontier = [root]
    for _ in range(depth):
        nxt: list[int] = []
        for rgn in frontier:
            ls = list(b.leaf_vars[rgn])
            rs.shuffle(ls)
            cut = int(np.round(0.5 * len(ls)))
            parts = [p for p in (ls[:cut], ls[cut:]) if p]
            if len(parts) == 1:
                continue
            ptn = new(False, b.scope[rgn])
            kids = [new(True, p) for p in parts]
            b.ins.setdefault(rgn, []).append(ptn)
            b.ins[ptn] = kids
            nxt.extend(kids)
        frontier = nxt
    return b.graph(root)


def fully_factorized(num_variables: int, *, num_repetitions: int = 1) -> RegionGraph:
    """The root region split `num_repetitions` times into all its single variables
    (templates/region_graph/algorithms/factorized.py:9-43); every repetition has its own leaf regions."""
    if num_variables <= 0:
        raise ValueError("The number of variables must be positive")
    if num_repetitions <= 0:
        raise ValueError("The number of repetitions must be positive")
    b = _RGBuilder()
    root = b.new(True, range(num_variables))
    b.ins[root] = []
    if num_variables == 1:
        return b.graph(root)
    for _ in range(num_repetitions):
        leaves = [b.new(True, [v]) for v in range(num_variables)]
        ptn = b.new(False, range(num_variables))
        b.ins[ptn] = leaves
        b.ins[root].append(ptn)
    return b.graph(root)


def linear_tree(num_variables: int, *, num_repetitions: int = 1, ordering: list[int] | None = None,
                randomize: bool = False, seed: int = 42) -> RegionGraph:
    """A chain of partitions that split off one variable at a time in the given (or, per repetition, shuffled)
    order (templates/region_graph/algorithms/linear.py:15-77)."""
    if num_variables <= 0:
        raise ValueError("The number of variables must be positive")
    if num_repetitions <= 0:
        raise ValueError("The number of repetitions must be positive")
    if ordering is not None and sorted(ordering) != list(range(num_variables)):
        raise ValueError(f"The variables ordering must be a permutation of values from 0 to {num_variables-1}")
    b = _RGBuilder()
    root = b.new(True, range(num_variables))
    if num_variables == 1:
        return b.graph(root)
    order = list(range(num_variables)) if ordering is None else list(ordering)
    rs = np.random.RandomState(seed) if randomize else None
    for _ in range(num_repetitions):
        if rs is not None:
            rs.shuffle(order)
        node, rest = root, set(range(num_variables))
        for v in order[:-1]:
            ptn = b.new(False, rest)
            rest = rest - {v}
            leaf = b.new(True, [v])
            nxt = b.new(True, rest)
            b.ins.setdefault(node, []).append(ptn)
            b.ins[ptn] = [leaf, nxt]
            node = nxt
    return b.graph(root)


# ---------------------------------------------------------------------------------------------
# unfolded layers and their parameter graphs
# ---------------------------------------------------------------------------------------------
@dataclass(eq=False)
class _PN:
    """One node of an unfolded parameter graph."""

    op: str
    shape: tuple[int, ...]
    cfg: dict = field(default_factory=dict)


@dataclass(eq=False)
class _PG:
    nodes: list[_PN]
    ins: dict[_PN, list[_PN]]
    output: _PN

    @property
    def shape(self) -> tuple[int, ...]:
        return self.output.shape


def _activation_node(shape: tuple[int, ...], activation: str) -> _PN | None:
    if activation == "softmax":
        return _PN("softmax", shape, {"dim": len(shape) - 1})
    if activation == "scaled-sigmoid":
        return _PN("scaled_sigmoid", shape, {"vmin": 1e-05, "vmax": 1.0})
    if activation == "sigmoid":
        return _PN("sigmoid", shape, {})
    if activation == "positive-clamp":  # (templates/utils.py:189-192: ClampParameter with vmin = 1e-18)
        return _PN("clamp", shape, {"vmin": 1e-18})
    if activation == "softplus":  # (templates/utils.py:193-194)
        return _PN("softplus", shape, {})
    if activation == "none":
        return None
    raise NotImplementedError(f"activation {activation!r}")


def _pg_tensor(shape: tuple[int, ...], activation: str) -> _PG:
    t = _PN("tensor", tuple(shape))
    a = _activation_node(tuple(shape), activation)
    if a is None:
        return _PG([t], {}, t)
    return _PG([t, a], {a: [t]}, a)


def _pg_mixing(num_units: int, arity: int, activation: str) -> _PG:
    """(K, H) mixing coefficients expanded to the (K, H*K) weight of a sum layer that takes a
    linear combination of its H input vectors (symbolic/parameters.py:1007-1044)."""
    pg = _pg_tensor((num_units, arity), activation)
    m = _PN("mixing_weight", (num_units, num_units * arity))
    return _PG([*pg.nodes, m], {**pg.ins, m: [pg.output]}, m)


def _pg_matmul(a: _PG, c: _PG) -> _PG:
    m = _PN("matmul", (a.shape[0], c.shape[1]))
    return _PG([*a.nodes, *c.nodes, m], {**a.ins, **c.ins, m: [a.output, c.output]}, m)


@dataclass
class InputSpec:
    """Input-layer family: ``categorical`` (probs = softmax over categories), ``embedding``
    (weight, no activation), ``gaussian`` (mean, stddev = scaled sigmoid) or ``binomial`` (probs = sigmoid;
    `num_states` values 0 .. total_count)."""

    name: str = "categorical"
    num_states: int = 256


@dataclass(eq=False)
class _L:
    kind: str  # 'input' | 'sum' | 'hadamard' | 'kronecker' | 'cpt' | 'tucker'
    ki: int
    ko: int
    arity: int = 1
    var: int = -1  # input layers
    spec: InputSpec | None = None  # input layers
    params: dict[str, _PG] = field(default_factory=dict)

    @property
    def type_name(self) -> str:
        return self.spec.name if self.kind == "input" else self.kind

    @property
    def config(self) -> dict[str, int]:
        if self.kind == "input":
            assert self.spec is not None
            cfg = {"num_output_units": self.ko}
            if self.spec.name == "categorical":
                cfg["num_categories"] = self.spec.num_states
            elif self.spec.name == "embedding":
                cfg["num_states"] = self.spec.num_states
            elif self.spec.name == "binomial":
                cfg["total_count"] = self.spec.num_states - 1
            return cfg
        if self.kind in ("hadamard", "kronecker"):
            return {"num_input_units": self.ki, "arity": self.arity}
        return {"num_input_units": self.ki, "num_output_units": self.ko, "arity": self.arity}

    def fold_key(self) -> tuple:
        """What decides whether two layers of a frontier are stacked: class, configuration and
        parameter shapes (layers/inner.py:45-47, layers/input.py:78-80) -- not the parameter graphs."""
        return (self.type_name, tuple(self.config.items()), tuple((n, p.shape) for n, p in self.params.items()))


def _input_layer(var: int, spec: InputSpec, k: int, activation: str | None) -> _L:
    if spec.name == "categorical":
        act = activation or "softmax"
        params = {("probs" if act == "softmax" else "logits"): _pg_tensor((k, spec.num_states), act)}
    elif spec.name == "embedding":
        params = {"weight": _pg_tensor((k, spec.num_states), activation or "none")}
    elif spec.name == "gaussian":
        params = {"mean": _pg_tensor((k,), "none"), "stddev": _pg_tensor((k,), "scaled-sigmoid")}
    elif spec.name == "binomial":  # probs = sigmoid(tensor), symbolic/layers.py:406-410
        params = {"probs": _pg_tensor((k,), "sigmoid")}
    else:
        raise NotImplementedError(f"input layer {spec.name!r} (supported: categorical, binomial, embedding, gaussian)")
    return _L("input", 1, k, var=var, spec=spec, params=params)


def _layers_of_region_graph(rg: RegionGraph, inputs, sum_product: str, k_in: int, k_sum: int, num_classes: int,
                            sum_activation: str, input_activation: str | None, use_mixing_weights: bool):
    """One input layer per variable, one sum-product block per partitioning, one mixing layer per
    region with several partitionings, bottom-up over the region graph."""
    if sum_product not in ("cp", "cp-t", "tucker"):
        raise NotImplementedError(f"Unknown sum-product layer abstraction called {sum_product}")
    layers: list[_L] = []
    ins: dict[_L, list[_L]] = {}
    of: dict[int, _L] = {}
    outs = _outgoings(rg.nodes, rg.ins)

    def spec_of(var: int) -> InputSpec:
        return inputs[var] if isinstance(inputs, (list, tuple)) else inputs

    def dense(ki: int, ko: int) -> _L:
        return _L("sum", ki, ko, params={"weight": _pg_tensor((ko, ki), sum_activation)})

    def same_units(kids: list[_L], what: str) -> int:
        units = {c.ko for c in kids}
        if len(units) > 1:
            raise ValueError(f"Cannot build a {what} layer, as the inputs would have different units")
        return units.pop()

    def block(node: int, kids: list[_L]) -> _L:
        is_root = not outs.get(node)
        if sum_product == "cp":
            denses = [dense(c.ko, k_sum) for c in kids]
            had = _L("hadamard", k_sum, k_sum, arity=len(kids))
            layers.extend(denses)
            layers.append(had)
            ins[had] = denses
            for d, c in zip(denses, kids):
                ins[d] = [c]
            if not is_root:
                return had
            od = dense(k_sum, num_classes)  # the output layer of the circuit must be a sum
            layers.append(od)
            ins[od] = [had]
            return od
        if sum_product == "cp-t":
            ku = same_units(kids, "CP transposed")
            prod = _L("hadamard", ku, ku, arity=len(kids))
        else:
            ku = same_units(kids, "Tucker")
            prod = _L("kronecker", ku, ku ** len(kids), arity=len(kids))
        d = dense(prod.ko, num_classes if is_root else k_sum)
        layers.extend([prod, d])
        ins[prod] = list(kids)
        ins[d] = [prod]
        return d

    for node in _kahn(rg.nodes, rg.ins):
        if not rg.is_region[node]:
            continue
        parts = rg.ins.get(node, [])
        if not parts:
            vs = rg.variables_of_leaf(node)
            if len(vs) > 1:  # fully factorised multivariate input
                leaves = [_input_layer(v, spec_of(v), k_in, input_activation) for v in vs]
                had = _L("hadamard", k_in, k_in, arity=len(leaves))
                layers.extend(leaves)
                ins[had] = leaves
                layers.append(had)
                of[node] = had
            else:
                of[node] = _input_layer(vs[0], spec_of(vs[0]), k_in, input_activation)
                layers.append(of[node])
        elif len(parts) == 1:
            of[node] = block(node, [of[r] for r in rg.ins[parts[0]]])
        else:
            units = k_sum if outs.get(node) else num_classes
            mix_ins = [block(node, [of[r] for r in rg.ins[p]]) for p in parts]
            weight = (_pg_mixing(units, len(mix_ins), sum_activation) if use_mixing_weights
                      else _pg_tensor((units, units * len(mix_ins)), sum_activation))
            mix = _L("sum", units, units, arity=len(mix_ins), params={"weight": weight})
            layers.append(mix)
            ins[mix] = mix_ins
            of[node] = mix
    return layers, ins, of[rg.root]


# ---------------------------------------------------------------------------------------------
# fusion rounds
# ---------------------------------------------------------------------------------------------
@dataclass(eq=False)
class _Match:
    pattern: str
    entries: list[_L]  # [the arity-1 sum, its single producer]


# pattern name -> kind of the producer feeding an arity-1 dense sum; registry order matters when a
# layer takes part in several matches (optimization/layers.py:468-472)
_FUSE_PATTERNS = (("sum-collapse", "sum"), ("tucker", "kronecker"), ("candecomp", "hadamard"))


def _apply_match(m: _Match) -> _L:
    top, src = m.entries
    if m.pattern == "sum-collapse":  # W_top (W_src x) = (W_top W_src) x
        return _L("sum", src.ki, top.ko, arity=src.arity,
                  params={"weight": _pg_matmul(top.params["weight"], src.params["weight"])})
    kind = "tucker" if m.pattern == "tucker" else "cpt"
    return _L(kind, src.ki, top.ko, arity=src.arity, params={"weight": top.params["weight"]})


def _fuse_round(layers: list[_L], ins: dict[_L, list[_L]], output: _L):
    """One pass of the reference's graph optimiser with its three fusion patterns; returns None when
    nothing matched."""
    order = _kahn(layers, ins)
    outs = _outgoings(layers, ins)
    found: dict[_L, list[_Match]] = {l: [] for l in order}
    for pattern, src_kind in _FUSE_PATTERNS:
        for l in order:
            if l.kind != "sum" or l.arity != 1:
                continue
            (src,) = ins[l]
            if src.kind != src_kind or len(outs.get(src, ())) > 1:
                continue
            m = _Match(pattern, [l, src])
            found[l].append(m)
            found[src].append(m)
    # a layer claimed by several matches keeps one: an already chosen match first, else the first found
    # (all matches have two entries, so "largest" is a stable tie)
    chosen: dict[_L, _Match] = {}
    for l in reversed(order):
        ms = found[l]
        if not ms:
            continue
        if len(ms) == 1:
            keep, drop = ms[0], []
        else:
            taken = list(chosen.values())
            keep = next((m for m in ms if any(m is t for t in taken)), None)
            if keep is None:
                keep, drop = ms[0], ms[1:]
            else:
                drop = [m for m in ms if m is not keep]
        chosen[l] = keep
        for m in drop:
            for e in m.entries:
                found[e].remove(m)
    if not chosen:
        return None
    fused = {id(m): _apply_match(m) for m in chosen.values()}
    new_layers: list[_L] = []
    new_ins: dict[_L, list[_L]] = {}
    entry_point: dict[int, _L] = {}

    def exit_of(l: _L) -> _L:
        return fused[id(chosen[l])] if l in chosen else l

    for l in order:
        m = chosen.get(l)
        if m is None:
            new_layers.append(l)
            new_ins[l] = [exit_of(c) for c in ins.get(l, ())]
            continue
        entry_point.setdefault(id(m), l)
        if l is m.entries[0]:  # the fused layer takes the place of the sum
            f = fused[id(m)]
            new_layers.append(f)
            new_ins[f] = [exit_of(c) for c in ins.get(entry_point[id(m)], ())]
    return new_layers, new_ins, exit_of(output)


def _optimize_layers(layers: list[_L], ins: dict[_L, list[_L]], output: _L, *, max_steps: int = 5):
    for _ in range(max_steps):
        res = _fuse_round(layers, ins, output)
        if res is None:
            break
        layers, ins, output = res
    return layers, ins, output


# ---------------------------------------------------------------------------------------------
# folding
# ---------------------------------------------------------------------------------------------
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


def _operand_index(column: list[tuple[int, int]], folds: dict[int, int]) -> FoldIndex:
    """Per-operand address-book entry of a parameter node (graph/folding.py:246-298): the folds of
    ONE operand, gathered from the concatenation of its distinct producers; the identity is free."""
    ids = list(dict.fromkeys(p for p, _ in column))
    base = dict(zip(ids, np.concatenate([[0], np.cumsum([folds[i] for i in ids])[:-1]]).astype(int)))
    cum = [int(base[p]) + f for p, f in column]
    if len(ids) == 1 and cum == list(range(folds[ids[0]])):
        return FoldIndex(ids, IDX_NONE)
    return FoldIndex(ids, IDX_ARRAY, np.asarray(cum, dtype=np.int64))


def _fold_param_graphs(pgs: list[_PG], new_tensor) -> ParamGraph:
    """Stack the parameter graphs of the layers of one fold group.  The graphs need not be alike
    (a collapsed sum and a plain mixing layer may share a group): nodes are merged frontier by
    frontier and grouped by operator, shape and configuration (torch/compiler.py:409-506)."""
    ins: dict[_PN, list[_PN]] = {}
    fronts: list[list[_PN]] = []
    for pg in pgs:
        ins.update(pg.ins)
        for i, fr in enumerate(_kahn_frontiers(pg.nodes, pg.ins)):
            if i < len(fronts):
                fronts[i].extend(fr)
            else:
                fronts.append(list(fr))
    nodes: list[ParamNode] = []
    folds: dict[int, int] = {}
    where: dict[_PN, tuple[int, int]] = {}
    for fr in fronts:
        groups: dict[tuple, list[_PN]] = {}
        for n in fr:
            key = (n.op, n.shape, tuple(sorted(n.cfg.items())), tuple(i.shape for i in ins.get(n, ())))
            groups.setdefault(key, []).append(n)
        for group in groups.values():
            mid = len(nodes)
            F = len(group)
            first = group[0]
            operands = [ins.get(n, []) for n in group]
            inputs = [_operand_index([where[ops[j]] for ops in operands], folds) for j in range(len(operands[0]))]
            for i, n in enumerate(group):
                where[n] = (mid, i)
            folds[mid] = F
            cfg = dict(first.cfg)
            if first.op == "tensor":
                cfg["tensor"] = new_tensor((F, *first.shape))
            nodes.append(ParamNode(first.op, F, first.shape, cfg, inputs))
    output = _fold_index([[where[pg.output] for pg in pgs]], folds, output=True)
    return ParamGraph(nodes, output, len(pgs), pgs[0].shape)


def _fold_layers(layers: list[_L], ins: dict[_L, list[_L]], output: _L, semiring: str, nvars: int, name: str) -> Plan:
    specs: list[LayerSpec] = []
    folds: dict[int, int] = {}
    where: dict[_L, tuple[int, int]] = {}
    tensors: dict[str, tuple[tuple[int, ...], str]] = {}

    def new_tensor(shape) -> str:
        t = f"t{len(tensors)}"
        tensors[t] = (tuple(int(s) for s in shape), "float32")
        return t

    for frontier in _kahn_frontiers(layers, ins):
        groups: dict[tuple, list[_L]] = {}
        for l in frontier:
            groups.setdefault(l.fold_key(), []).append(l)
        for group in groups.values():
            first = group[0]
            F = len(group)
            mid = len(specs)
            rows = [[where[c] for c in ins.get(l, ())] for l in group]
            for i, l in enumerate(group):
                where[l] = (mid, i)
            folds[mid] = F
            params = {pn: _fold_param_graphs([l.params[pn] for l in group], new_tensor) for pn in first.params}
            if first.kind == "input":
                scope = np.asarray([[l.var] for l in group], dtype=np.int64)
                specs.append(LayerSpec(first.type_name, F, 1, 1, first.ko, first.config, params, None, scope))
            else:
                specs.append(LayerSpec(first.kind, F, first.arity, first.ki, first.ko, first.config, params,
                                       _fold_index(rows, folds)))
    return Plan(semiring, nvars, specs, _fold_index([[where[output]]], folds, output=True), tensors, name)


def build_plan(
    rg: RegionGraph,
    *,
    input_layer: InputSpec | str | list = "categorical",
    sum_product: str = "cp",
    num_input_units: int = 32,
    num_sum_units: int = 32,
    num_classes: int = 1,
    sum_activation: str = "softmax",
    input_activation: str | None = None,
    use_mixing_weights: bool = True,
    semiring: str = "lse-sum",
    name: str = "",
) -> Plan:
    """Region graph -> the folded, optimised plan ``compile(fold=True, optimize=True)`` would give.
    `input_layer` may be a list with one `InputSpec` per variable."""
    inputs = InputSpec(input_layer) if isinstance(input_layer, str) else input_layer
    nvars = max(max(s) for s in rg.scope.values()) + 1
    if isinstance(inputs, (list, tuple)) and len(inputs) != nvars:
        raise ValueError(f"Number of provided input layers ({len(inputs)}) does not match the number of features ({nvars}).")
    layers, ins, out = _layers_of_region_graph(rg, inputs, sum_product, num_input_units, num_sum_units, num_classes,
                                               sum_activation, input_activation, use_mixing_weights)
    layers = _kahn(layers, ins)  # compile order (torch/compiler.py:275-282)
    layers, ins, out = _optimize_layers(layers, ins, out)
    return _fold_layers(layers, ins, out, semiring, nvars, name)


def quad_tree_plan(shape: tuple[int, int, int] = (1, 28, 28), **kw: Any) -> Plan:
    """``image_data(shape, 'quad-tree-2', ...)`` + ``compile(fold=True, optimize=True)``."""
    c, h, w = shape
    return build_plan(quad_tree(h, w, num_patch_splits=kw.pop("num_patch_splits", 2), num_channels=c), **kw)


def random_binary_tree_plan(num_features: int, *, depth: int | None = None, seed: int = 42, **kw: Any) -> Plan:
    """``tabular_data('random-binary-tree', num_features=...)`` + compile."""
    return build_plan(random_binary_tree(num_features, depth=depth, seed=seed), **kw)


# ---------------------------------------------------------------------------------------------
# the reference's template entry points (cirkit/templates/data_modalities.py:26-305), same argument
# names, returning the compiled plan instead of a symbolic circuit
# ---------------------------------------------------------------------------------------------
def image_data(
    image_shape: tuple[int, int, int],
    region_graph: str = "quad-graph",
    *,
    input_layer: str = "categorical",
    num_input_units: int,
    sum_product_layer: str = "cp",
    num_sum_units: int,
    num_classes: int = 1,
    sum_weight_activation: str = "softmax",
    input_activation: str | None = None,
    use_mixing_weights: bool = True,
    semiring: str = "lse-sum",
) -> Plan:
    """``data_modalities.image_data`` (:26-162) followed by ``compile(fold=True, optimize=True)``."""
    if not isinstance(image_shape, tuple) or len(image_shape) != 3 or any(d <= 0 for d in image_shape):
        raise ValueError(f"Expected the image shape to be a tuple of three positive integers, but found {image_shape}")
    if input_layer not in ("categorical", "binomial", "embedding", "gaussian"):
        raise ValueError(f"Unknown input layer called {input_layer}")
    c, h, w = image_shape
    if region_graph == "quad-tree-2":
        rg = quad_tree(h, w, num_patch_splits=2, num_channels=c)
    elif region_graph == "quad-tree-4":
        rg = quad_tree(h, w, num_patch_splits=4, num_channels=c)
    elif region_graph == "quad-graph":
        rg = quad_graph(h, w, num_channels=c)
    elif region_graph == "random-binary-tree":
        rg = random_binary_tree(c * h * w)
    elif region_graph == "poon-domingos":
        rg = poon_domingos(image_shape, delta=int(max(np.ceil(h / 8), np.ceil(w / 8))))
    else:
        raise ValueError(f"Unknown region graph called {region_graph}")
    states = 256  # data_modalities.py:120-126: categorical / embedding over 8-bit pixels
    return build_plan(
        rg, input_layer=InputSpec(input_layer, states), sum_product=sum_product_layer,
        num_input_units=num_input_units, num_sum_units=num_sum_units, num_classes=num_classes,
        sum_activation=sum_weight_activation, input_activation=input_activation,
        use_mixing_weights=use_mixing_weights, semiring=semiring,
    )


def _tabular_input_spec(d: dict) -> InputSpec:
    args = dict(d.get("args", {}))
    if d["name"] == "binomial":
        return InputSpec("binomial", int(args.get("total_count", 1)) + 1)
    return InputSpec(d["name"], int(args.get("num_categories", args.get("num_states", 2))))


def tabular_data(
    region_graph: str = "random-binary-tree",
    *,
    data=None,
    num_features: int | None = None,
    input_layers: dict | list,
    num_input_units: int,
    sum_product_layer: str = "cp",
    num_sum_units: int,
    num_classes: int = 1,
    use_mixing_weights: bool = True,
    semiring: str = "lse-sum",
) -> Plan:
    """``data_modalities.tabular_data`` (:165-305) followed by ``compile(fold=True, optimize=True)``.
    `input_layers` is one ``{'name': ..., 'args': {...}}`` dict or a list with one per feature."""
    if region_graph == "chow-liu-tree":
        raise NotImplementedError("structure learning (region_graph='chow-liu-tree') is outside this backend's scope: compile the "
                                  "circuit with cirkit and pass the compiled TorchCircuit to cirkit_amd.pipeline.compile")
    if region_graph == "random-binary-tree":
        if num_features is None:
            if data is None:
                raise ValueError(f"You must pass `num_features=` if you ask for {region_graph}.")
            num_features = int(data.shape[1])
        rg = random_binary_tree(num_features)
    else:
        raise ValueError(f"Unknown region graph called {region_graph}")
    specs = _tabular_input_spec(input_layers) if isinstance(input_layers, dict) else [_tabular_input_spec(d) for d in input_layers]
    return build_plan(
        rg, input_layer=specs, sum_product=sum_product_layer,
        num_input_units=num_input_units, num_sum_units=num_sum_units, num_classes=num_classes,
        use_mixing_weights=use_mixing_weights, semiring=semiring,
    )
