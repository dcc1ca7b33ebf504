"""Pad the unit counts of a plan to multiples of 32.

The matrix-core kernels of this backend work on tiles of 32 units (cirkit_amd/csrc/ck_tile.h); a layer
whose unit counts are not multiples of 32 runs on the shape-generic kernels, 2-5x slower than a tile with
idle columns would be.  `pad_units` rewrites a plan so that every unit count K > 1 becomes the next multiple
of 32 WITHOUT changing what the circuit computes:

* the raw parameter tensors are enlarged; entries of a sum weight that multiply a padded INPUT unit are
  filled so that the parameter graph maps them to exactly 0 (``-inf`` under softmax / exp / sigmoid, ``0``
  for an unconstrained weight), so a padded unit never contributes to a real one;
* every padded unit -- of an input layer or of a sum layer -- is a COPY of a real unit of the same layer (same
  parameters: copied input distributions, copied weight rows), so at every layer the padded values repeat real
  values and the row maxima of the log-sum-exp reductions (semiring.py:383-408) stay where they were.  A dummy unit
  could be astronomically more likely than every real unit at some input (a Gaussian in its tail, a Binomial at the
  end of its support, a uniform mixture next to outputs that all avoid the one large input) and the shifted
  exponentials of all real units would underflow; nobody reads the copies (weight 0 on every padded input);
* the parameter graphs themselves are unchanged (same nodes, new shapes), so training updates keep the
  invariant: the gradient of a softmax w.r.t. a ``-inf`` logit is 0.

Real (lse-sum) circuits made of Categorical / Gaussian / Sum / CP-T / Tucker / Hadamard layers with
tensor, softmax, sigmoid, scaled-sigmoid, exp, mixing-weight and matmul parameter nodes are covered --
everything `cirkit_amd.templates` builds for them; anything else is left as it is (`pad_units` returns
None and the circuit runs unpadded).
"""

from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from typing import Mapping

import numpy as np

from .plan import LayerSpec, ParamGraph, ParamNode, Plan

_UNARY = {"softmax", "sigmoid", "scaled_sigmoid", "exp", "square", "softplus"}
_ZERO_AT_NEG_INF = {"softmax", "sigmoid", "exp", "softplus"}  # ops that map -inf to exactly 0 (a positive clamp has no zero: not padded)
_LAYERS = {"categorical", "binomial", "gaussian", "sum", "cpt", "tucker", "hadamard"}


@dataclass(frozen=True)
class _Role:
    """One axis of a parameter: `reps` groups of `k` units, concatenated ("cat") or as the flattened
    index of a `reps`-fold product ("kron"); kind "in" = multiplies input units (padded entries evaluate to 0),
    "dup" = produces units (padded entries are copies of real ones)."""

    k: int
    reps: int
    mode: str
    kind: str

    def size(self, k: int) -> int:
        return self.reps * k if self.mode == "cat" else k**self.reps

    def positions(self, kp: int) -> np.ndarray:
        """Index of every real entry inside the padded axis."""
        if self.mode == "cat":
            return (np.arange(self.reps)[:, None] * kp + np.arange(self.k)[None, :]).reshape(-1)
        grid = np.indices((self.k,) * self.reps).reshape(self.reps, -1)
        return np.ravel_multi_index(tuple(grid), (kp,) * self.reps)


@dataclass
class PadInfo:
    """What `pad_units` did: how to move parameter values between the two shapes."""

    multiple: int
    tensors: dict[str, tuple[tuple, float, float]] = field(default_factory=dict)  # name -> (roles, fill_in, fill_out)
    shapes: dict[str, tuple[tuple[int, ...], tuple[int, ...]]] = field(default_factory=dict)  # name -> (old, new)
    out_units: int = 1  # unit count of the circuit output before padding

    def pad(self, name: str, value: np.ndarray) -> np.ndarray:
        roles, fill_in, fill_out = self.tensors[name]
        old, new = self.shapes[name]
        value = np.asarray(value)
        if tuple(value.shape) != tuple(old):
            raise ValueError(f"tensor {name!r}: expected shape {old}, found {tuple(value.shape)}")
        out = np.full(new, fill_out, dtype=value.dtype)
        pos = []
        for ax, r in enumerate(roles, start=1):  # axis 0 = folds
            if r is None:
                pos.append(np.arange(old[ax]))
                continue
            p = r.positions(_pad(r.k, self.multiple))
            pos.append(p)
            if r.kind == "in":
                mask = np.ones(new[ax], dtype=bool)
                mask[p] = False
                sl = [slice(None)] * len(new)
                sl[ax] = mask
                out[tuple(sl)] = fill_in
        out[np.ix_(np.arange(old[0]), *pos)] = value
        for ax, r in enumerate(roles, start=1):
            if r is not None and r.kind == "dup" and new[ax] > old[ax]:  # padded units repeat the real ones cyclically
                dst = [slice(None)] * len(new)
                dst[ax] = slice(old[ax], new[ax])
                out[tuple(dst)] = np.take(out, np.arange(old[ax], new[ax]) % old[ax], axis=ax)
        return out

    def duplicated_axes(self, name: str) -> list[tuple[int, int, int]]:
        """(axis, real size, padded size) of the axes of tensor `name` whose padded entries are copies of real ones
        (input-layer units) -- whoever updates the real entries in place must refresh the copies."""
        roles, _, _ = self.tensors[name]
        old, new = self.shapes[name]
        return [(ax, old[ax], new[ax]) for ax, r in enumerate(roles, start=1)
                if r is not None and r.kind == "dup" and new[ax] > old[ax]]

    def unpad(self, name: str, value: np.ndarray) -> np.ndarray:
        roles, _, _ = self.tensors[name]
        old, _ = self.shapes[name]
        pos = [np.arange(old[ax]) if r is None else r.positions(_pad(r.k, self.multiple))
               for ax, r in enumerate(roles, start=1)]
        return np.asarray(value)[np.ix_(np.arange(old[0]), *pos)]


def _pad(k: int, multiple: int) -> int:
    return k if k <= 1 else multiple * math.ceil(k / multiple)


def _param_roles(layer: LayerSpec, name: str) -> list[_Role | None] | None:
    """Unit structure of the axes of a layer parameter (per-fold shape)."""
    ki, ko, h = layer.num_input_units, layer.num_output_units, layer.arity
    # input layers: a padded unit is a COPY of a real unit (kind "dup"), so its value never exceeds the largest real
    # value of its row -- a dummy distribution could, and the row maximum of the first log-sum-exp would then underflow
    # every real unit (e.g. a Gaussian evaluated far in its tail, a Binomial at the end of its support)
    if layer.type in ("categorical",) and name in ("probs", "logits"):
        return [_Role(ko, 1, "cat", "dup"), None]
    if layer.type == "gaussian" and name in ("mean", "stddev", "log_partition"):
        return [_Role(ko, 1, "cat", "dup")]
    if layer.type == "binomial" and name in ("probs", "logits"):
        return [_Role(ko, 1, "cat", "dup")]
    if layer.type == "sum" and name == "weight":
        return [_Role(ko, 1, "cat", "dup"), _Role(ki, h, "cat", "in")]
    if layer.type == "cpt" and name == "weight":
        return [_Role(ko, 1, "cat", "dup"), _Role(ki, 1, "cat", "in")]
    if layer.type == "tucker" and name == "weight" and h == 2:  # (K^H inputs: padding a higher arity explodes)
        return [_Role(ko, 1, "cat", "dup"), _Role(ki, h, "kron", "in")]
    return None


class _Unsupported(Exception):
    pass


def _pad_graph(g: ParamGraph, roles: list[_Role | None], multiple: int, info: PadInfo, plan: Plan,
               positive: bool = False) -> ParamGraph:
    """New shapes for every node of a parameter graph; registers the leaf tensors in `info`.
    `positive`: the parameter must stay positive on padded units (a Gaussian's stddev)."""
    # fold gathers (the output and the operands may pick / concatenate folds of several nodes) do not touch the
    # unit axes: every node behind a gather gets the roles of the operand
    node_roles: dict[int, list[_Role | None]] = {}
    above: dict[int, str | None] = {}  # the op that consumes the node

    def assign(ids, r, op):
        for i in ids:
            if i in node_roles:
                if node_roles[i] != r:
                    raise _Unsupported("a parameter node is consumed with two unit structures")
                if above[i] != op:  # e.g. mixing weights used as they are by some folds and inside a product by others
                    above[i] = "*"  # (only matters for a raw tensor, which is then refused)
                continue
            node_roles[i] = r
            above[i] = op

    assign(g.output.ids, roles, None)
    new_nodes = copy.deepcopy(g.nodes)
    for i in range(len(g.nodes) - 1, -1, -1):
        n = g.nodes[i]
        if i not in node_roles:
            raise _Unsupported("parameter node without a consumer")
        r = node_roles[i]
        if len(r) != len(n.shape):
            raise _Unsupported("rank mismatch")
        for ax, role in enumerate(r):
            if role is not None and role.size(role.k) != n.shape[ax]:
                raise _Unsupported("axis size does not match the unit structure")
        new_nodes[i].shape = tuple(s if role is None else role.size(_pad(role.k, multiple)) for s, role in zip(n.shape, r))
        ins = [fi.ids for fi in n.inputs]
        if n.op == "tensor":
            name = n.config["tensor"]
            if any(role is not None and role.kind == "in" for role in r):
                up = above[i]
                if up is None:
                    fill_in = 0.0
                elif up in _ZERO_AT_NEG_INF:
                    fill_in = -math.inf
                else:
                    raise _Unsupported(f"no zero pre-image under {up!r}")
                if up == "softmax":
                    d = _consumer(g, i).config.get("dim")
                    in_axes = [ax for ax, role in enumerate(r) if role is not None and role.kind == "in"]
                    if in_axes != [d]:
                        raise _Unsupported("softmax does not run over the input units")
            else:
                fill_in = 0.0
            old = tuple(plan.tensors[name][0])
            new = (old[0], *new_nodes[i].shape)
            fill_out = 1.0 if positive and above[i] in (None, "square") else 0.0
            entry = (tuple(r), fill_in, fill_out)
            if name in info.tensors and (info.tensors[name] != entry or info.shapes[name] != (old, new)):
                raise _Unsupported("a tensor is used with two different unit structures")
            info.tensors[name] = entry
            info.shapes[name] = (old, new)
        elif n.op in _UNARY:
            if n.op == "softmax" and any(role is not None and role.kind == "dup" and ax == n.config.get("dim")
                                         for ax, role in enumerate(r)):
                raise _Unsupported("softmax over output units")
            assign(ins[0], r, n.op)
        elif n.op == "mixing_weight":
            # (K, H) -> (K, H K): w[k, h K + k'] = m[k, h] delta(k, k')   (nodes.py:847-862)
            k = n.shape[0]
            if r[1] is not None and (r[1].mode != "cat" or r[1].k != k):
                raise _Unsupported("mixing weight with another unit structure")
            row = r[0] if r[0] is not None else (_Role(k, 1, "cat", "dup") if r[1] is not None else None)
            if row is not None and r[1] is None and k > 1:
                raise _Unsupported("mixing weight rows padded without its columns")
            new_nodes[i].shape = (row.size(_pad(k, multiple)) if row is not None else k, new_nodes[i].shape[1])
            assign(ins[0], [None if row is None else _Role(k, 1, "cat", "dup"), None], n.op)
        elif n.op == "matmul":
            a, b = ins
            km = g.nodes[a[0]].shape[1]
            inner_in = _Role(km, 1, "cat", "in") if _pad(km, multiple) != km else None
            inner_out = _Role(km, 1, "cat", "dup") if inner_in is not None else None
            assign(a, [r[0], inner_in], n.op)
            assign(b, [inner_out, r[1]], n.op)
        else:
            raise _Unsupported(f"parameter op {n.op!r}")
    return ParamGraph(new_nodes, copy.deepcopy(g.output), g.num_folds, tuple(new_nodes[g.output.ids[0]].shape))


def _consumer(g: ParamGraph, i: int) -> ParamNode:
    for n in g.nodes:
        if any(i in fi.ids for fi in n.inputs):
            return n
    raise _Unsupported("no consumer")


def pad_units(plan: Plan, multiple: int = 32) -> tuple[Plan, PadInfo] | None:
    """The plan with every unit count K > 1 rounded up to a multiple of `multiple`, and the record needed
    to convert parameter values (`PadInfo.pad` / `PadInfo.unpad`).  None when nothing changes or when the
    plan contains something the transformation does not cover."""
    if plan.semiring != "lse-sum":
        return None
    units = {l.num_input_units for l in plan.layers} | {l.num_output_units for l in plan.layers}
    if all(_pad(k, multiple) == k for k in units):
        return None
    if any(l.type not in _LAYERS for l in plan.layers):
        return None
    info = PadInfo(multiple)
    layers = []
    try:
        for l in plan.layers:
            params = {}
            for name, g in l.params.items():
                roles = _param_roles(l, name)
                if roles is None:
                    raise _Unsupported(f"{l.type}.{name}")
                params[name] = _pad_graph(g, roles, multiple, info, plan, positive=(l.type, name) == ("gaussian", "stddev"))
            ki, ko = _pad(l.num_input_units, multiple), _pad(l.num_output_units, multiple)
            cfg = dict(l.config)
            if "num_input_units" in cfg:
                cfg["num_input_units"] = ki
            if "num_output_units" in cfg:
                cfg["num_output_units"] = ko
            layers.append(LayerSpec(l.type, l.num_folds, l.arity, ki, ko, cfg, params, copy.deepcopy(l.inputs),
                                    None if l.scope_idx is None else l.scope_idx.copy()))
    except _Unsupported:
        return None
    tensors = {k: (info.shapes[k][1] if k in info.shapes else tuple(s), dt) for k, (s, dt) in plan.tensors.items()}
    for k, (s, _) in plan.tensors.items():  # tensors that no padded axis touches keep their shape
        if k not in info.shapes:
            info.shapes[k] = (tuple(s), tuple(s))
            info.tensors[k] = (tuple([None] * (len(s) - 1)), 0.0, 0.0)
    out_layers = {i for i in plan.output.ids}
    info.out_units = max(plan.layers[i].num_output_units for i in out_layers)
    return Plan(plan.semiring, plan.num_variables, layers, copy.deepcopy(plan.output), tensors, plan.name), info


def pad_tensors(info: PadInfo, tensors: Mapping[str, object]) -> dict[str, np.ndarray]:
    """Parameter values of the original plan -> values of the padded plan."""
    out = {}
    for name, v in tensors.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        out[name] = info.pad(name, a) if name in info.tensors else a
    return out
