"""Plan analysis for cross-layer fusion (host side, once per circuit).

Finds the leaf region  Categorical -> [dense Sum] -> CP-T -> CP-T ...  in which every fold of a
layer is consumed exactly once, by the next layer only -- i.e. a forest of complete binary trees
over the input folds -- and flattens it into the node tables `ck_subtree_cat_cpt_fwd` walks.

This is a property of the folded plan the reference builds (``build_folded_graph``,
cirkit/backend/torch/graph/folding.py:62-166): region-graph templates such as QuadTree stack each
tree level into one folded layer, so consecutive folded layers are exactly such a forest until the
first level where a fold is shared or left over.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

MAX_DEPTH = 4
FUSED_K = 32


@dataclass
class SubtreeGroup:
    input_layer: int
    dense_layer: int | None
    levels: list[int]  # CP-T layers, level 1..D
    nodes: np.ndarray  # packed int32 tables
    node_off: list[int]  # per level 0..D
    leaf_off: int

    @property
    def root(self) -> int:
        return self.levels[-1] if self.levels else self.dense_layer

    @property
    def depth(self) -> int:
        return len(self.levels)

    @property
    def virtual(self) -> list[int]:
        """Layers whose outputs are never materialised."""
        chain = [self.input_layer] + ([self.dense_layer] if self.dense_layer is not None else []) + self.levels
        return chain[:-1]


def _consumers(children, out_pairs, n_layers):
    cons = [set() for _ in range(n_layers)]
    for j, ch in enumerate(children):
        if ch is not None:
            for p in np.unique(ch[..., 0]):
                cons[int(p)].add(j)
    for p in np.unique(out_pairs[:, 0]):
        cons[int(p)].add(-1)  # the circuit output
    return cons


def _uses_each_fold_once(ch: np.ndarray, producer: int, n_folds: int) -> bool:
    if not np.all(ch[..., 0] == producer):
        return False
    folds = np.sort(ch[..., 1].reshape(-1))
    return len(folds) == n_folds and np.array_equal(folds, np.arange(n_folds))


def find_subtree_groups(plan, layers, children, out_pairs, max_depth: int = MAX_DEPTH) -> list[SubtreeGroup]:
    """`layers`: the HipLayer objects; `children[j]`: (F_j, H_j, 2) producer/fold pairs or None."""
    if plan.semiring != "lse-sum":
        return []
    max_depth = max(0, min(int(max_depth), MAX_DEPTH))
    cons = _consumers(children, out_pairs, len(layers))
    groups: list[SubtreeGroup] = []
    for i0, (spec, l) in enumerate(zip(plan.layers, layers)):
        if spec.type != "categorical" or l.num_output_units != FUSED_K or spec.scope_idx.shape[1] != 1:
            continue
        cur, dense, levels = i0, None, []
        while len(cons[cur]) == 1:
            (j,) = cons[cur]
            if j < 0:
                break
            sj, lj, ch = plan.layers[j], layers[j], children[j]
            if lj.num_input_units != FUSED_K or lj.num_output_units != FUSED_K:
                break
            if not _uses_each_fold_once(ch, cur, layers[cur].num_folds):
                break
            if getattr(lj, "_mixing", False):
                break
            if sj.type == "sum" and lj.arity == 1 and cur == i0 and dense is None:
                dense = j
            elif sj.type == "cpt" and lj.arity == 2 and len(levels) < max_depth:
                levels.append(j)
            else:
                break
            cur = j
        if not levels and dense is None:
            continue
        D = len(levels)
        F_root = layers[cur].num_folds
        # node tables, top-down
        tabs = [None] * (D + 1)
        tabs[D] = np.arange(F_root, dtype=np.int64)[:, None]
        for lv in range(D, 0, -1):
            ch = children[levels[lv - 1]]  # (F_l, 2, 2)
            tabs[lv - 1] = ch[tabs[lv], :, 1].reshape(F_root, -1)
        if dense is not None:
            leaf = children[dense][tabs[0], 0, 1]  # dense fold -> input fold
        else:
            leaf = tabs[0]
        packed, node_off = [], []
        off = 0
        for tb in tabs:
            node_off.append(off)
            packed.append(tb.reshape(-1))
            off += tb.size
        leaf_off = off
        packed.append(np.asarray(leaf).reshape(-1))
        groups.append(
            SubtreeGroup(i0, dense, levels, np.concatenate(packed).astype(np.int32), node_off, leaf_off)
        )
    return groups


MAX_TAIL_LAYERS = 12


def find_tail(plan, layers, skip: set[int], max_folds: int = 64) -> list[int]:
    """Trailing layers with few folds that `ck_tail_lse_fwd` evaluates in one launch: real CP-T /
    dense sum steps with 32 input units, 32 output units (fewer only for terminal layers, e.g. the
    scalar root), at most `max_folds` folds each."""
    if plan.semiring != "lse-sum":
        return []
    tail: list[int] = []
    for i in range(len(layers) - 1, -1, -1):
        s, l = plan.layers[i], layers[i]
        ok = (
            i not in skip
            and s.inputs is not None
            and (s.type == "cpt" or (s.type == "sum" and l.arity == 1))
            and not getattr(l, "_mixing", False)
            and l.num_input_units == FUSED_K
            and (l.num_output_units == FUSED_K or (l.num_output_units < FUSED_K and not tail))
            and l.num_folds <= max_folds
            and len(tail) < MAX_TAIL_LAYERS
        )
        if not ok:
            break
        tail.append(i)
    tail.reverse()
    # a terminal layer with Ko < 32 may only be consumed by the circuit output
    return tail if len(tail) >= 2 else []
