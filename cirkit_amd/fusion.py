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


def find_subtree_groups(plan, layers, children, out_pairs, max_depth: int = MAX_DEPTH, *,
                        signed: bool = False) -> list[SubtreeGroup]:
    """`layers`: the HipLayer objects; `children[j]`: (F_j, H_j, 2) producer/fold pairs or None.
    `signed`: a real-valued circuit under complex-lse-sum -- the leaves are Embedding layers (signed table rows), the
    levels CP-T layers with plain real weights, no dense layer in between (ck_leaf_persistent_fwd with signed_redo)."""
    if plan.semiring != ("complex-lse-sum" if signed else "lse-sum"):
        return []
    max_depth = max(0, min(int(max_depth), MAX_DEPTH))
    cons = _consumers(children, out_pairs, len(layers))
    groups: list[SubtreeGroup] = []
    for i0, (spec, l) in enumerate(zip(plan.layers, layers)):
        if spec.type != ("embedding" if signed else "categorical") or l.num_output_units != FUSED_K or spec.scope_idx.shape[1] != 1:
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
            if sj.type == "sum" and lj.arity == 1 and cur == i0 and dense is None and not signed:
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


def find_tail(plan, layers, skip: set[int], max_folds: int = 64, *, signed: bool = False) -> list[int]:
    """Trailing layers with few folds that `ck_tail16_lse_fwd` / `ck_tail_params_fwd` evaluate in one launch: real CP-T /
    dense sum steps with 32 input units, 32 output units (fewer only for terminal layers, e.g. the
    scalar root), at most `max_folds` folds each."""
    if plan.semiring != ("complex-lse-sum" if signed else "lse-sum"):
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
    # the walk keeps every fold output of the tail in LDS (ck_tail16.hip: at most 64 folds, children of arity <= 4): a longer
    # candidate loses its first layers, which then run as ordinary layer launches
    while tail and (sum(layers[i].num_folds for i in tail) > 64 or layers[tail[0]].arity > 4):
        tail.pop(0)
    if any(layers[i].arity > 4 for i in tail):
        return []
    # a terminal layer with Ko < 32 may only be consumed by the circuit output
    return tail if len(tail) >= 2 else []


# ---------------------------------------------------------------------------------------------
# CP blocks: dense sum layers folded into the Hadamard layer that multiplies their outputs
# ---------------------------------------------------------------------------------------------
CP_K = (32, 64)


@dataclass
class CPBlock:
    """One Hadamard layer evaluated by `ck_cp_lse_fwd`: slot (f, s) reads `slot_child[f, s]`
    (producer layer, fold) and, where `slot_dense[f, s, 0] >= 0`, pushes it through fold
    `slot_dense[f, s, 1]` of dense layer `slot_dense[f, s, 0]` first."""

    layer: int
    slot_child: np.ndarray  # (F, S, 2)
    slot_dense: np.ndarray  # (F, S, 2), -1 = plain slot
    post: bool = False  # the consumer is a CP-T layer: its own dense sum follows the product


def find_cp_blocks(plan, layers, children, out_pairs, skip: set[int]):
    """Returns (blocks, leftover, virtual).

    A fold of a dense layer (arity-1 real sum, K -> K with K in CP_K, not a mixing layer) that is
    consumed exactly once, by a Hadamard or CP-T layer with the same K, is evaluated inside that
    consumer's launch and never written to memory.  `leftover[d]` lists the folds of dense layer d that other
    consumers still need (they are evaluated in place by a launch over that subset); a dense layer
    without leftovers is `virtual` (no activation storage at all)."""
    if plan.semiring != "lse-sum":
        return [], {}, set()
    n = len(layers)

    def is_dense(j: int) -> bool:
        s, l = plan.layers[j], layers[j]
        return (j not in skip and s.type == "sum" and l.arity == 1 and not getattr(l, "_mixing", False)
                and l.num_input_units == l.num_output_units and l.num_output_units in CP_K
                and not l.is_complex)

    def is_prod(j: int) -> bool:
        """A Hadamard layer, or a CP-T layer (Hadamard -> dense sum, K -> K) -- both multiply their children."""
        s, l = plan.layers[j], layers[j]
        if j in skip or l.num_input_units not in CP_K or l.is_complex:
            return False
        if s.type == "hadamard":
            return l.arity >= 2
        return s.type == "cpt" and l.num_output_units == l.num_input_units

    # how often each (layer, fold) is read, and by whom
    uses = [np.zeros(l.num_folds, dtype=np.int64) for l in layers]
    by_prod = [np.zeros(l.num_folds, dtype=np.int64) for l in layers]
    for j, ch in enumerate(children):
        if ch is None:
            continue
        flat = ch.reshape(-1, 2)
        for p in np.unique(flat[:, 0]):
            cnt = np.bincount(flat[flat[:, 0] == p, 1], minlength=layers[int(p)].num_folds)
            uses[int(p)] += cnt
            if is_prod(j) and layers[j].num_input_units == layers[int(p)].num_output_units:
                by_prod[int(p)] += cnt
    for p, f in out_pairs:
        uses[int(p)][int(f)] += 1
    fusable = {d: (uses[d] == 1) & (by_prod[d] == 1) for d in range(n) if is_dense(d)}
    fusable = {d: m for d, m in fusable.items() if m.any()}
    if not fusable:
        return [], {}, set()
    blocks: list[CPBlock] = []
    for j in range(n):
        if not is_prod(j):
            continue
        ch = children[j]  # (F, S, 2)
        slot_child = ch.copy()
        slot_dense = np.full_like(ch, -1)
        hit = False
        for d, mask in fusable.items():
            sel = (ch[..., 0] == d) & mask[np.where(ch[..., 0] == d, ch[..., 1], 0)]
            if not sel.any():
                continue
            hit = True
            folds = ch[..., 1][sel]
            slot_dense[sel] = np.stack([np.full_like(folds, d), folds], axis=-1)
            slot_child[sel] = children[d][folds, 0]  # the dense fold's own input
        if hit:
            blocks.append(CPBlock(j, slot_child, slot_dense, post=plan.layers[j].type == "cpt"))
    leftover = {d: np.nonzero(~m)[0] for d, m in fusable.items() if not m.all()}
    virtual = {d for d, m in fusable.items() if m.all()}
    return blocks, leftover, virtual


# ---------------------------------------------------------------------------------------------
# factorised multivariate inputs: Gaussian folds multiplied straight away by a Hadamard layer
# ---------------------------------------------------------------------------------------------
def find_input_products(plan, layers, children, out_pairs, skip: set[int]) -> dict[int, int]:
    """{hadamard layer: gaussian layer} where EVERY fold of the Gaussian layer is read exactly once,
    by that Hadamard layer only: `ck_gaussian_prod_fwd` then evaluates the product directly from
    the batch and the Gaussian layer output is never written."""
    if plan.semiring != "lse-sum":
        return {}
    found: dict[int, int] = {}
    outs = {int(p) for p in out_pairs[:, 0]}
    for j, (s, ch) in enumerate(zip(plan.layers, children)):
        if j in skip or s.type != "hadamard" or ch is None:
            continue
        prods = np.unique(ch[..., 0])
        if len(prods) != 1:
            continue
        g = int(prods[0])
        sg = plan.layers[g]
        if g in skip or g in outs or sg.type != "gaussian" or sg.scope_idx.shape[1] != 1:
            continue
        if any(c is not None and k != j and (c[..., 0] == g).any() for k, c in enumerate(children)):
            continue
        if not _uses_each_fold_once(ch, g, layers[g].num_folds):
            continue
        found[j] = g
    return found


# ---------------------------------------------------------------------------------------------
# regions: a mixing layer together with the CP blocks it combines
# ---------------------------------------------------------------------------------------------
@dataclass
class RegionBlock:
    """One mixing layer evaluated by `ck_region_lse_fwd`: partitioning (f, h) is the CP block
    `slot_child[f, h]` / `slot_dense[f, h]` (S slots, as in `CPBlock`)."""

    layer: int
    slot_child: np.ndarray  # (F, H, S, 2)
    slot_dense: np.ndarray  # (F, H, S, 2)


def find_region_blocks(plan, layers, children, out_pairs, cp_blocks: dict[int, CPBlock], skip: set[int]):
    """Returns (regions, absorbed).  A mixing layer whose every input is a fold of a CP-block
    Hadamard layer, read by nobody else, takes those folds over: `absorbed[hadamard layer]` is the
    boolean mask of its folds that are no longer evaluated (nor stored) on their own.  May add
    plain-slot blocks for bare Hadamard layers to `cp_blocks`."""
    if plan.semiring != "lse-sum":
        return [], {}
    uses = [np.zeros(l.num_folds, dtype=np.int64) for l in layers]
    for ch in children:
        if ch is not None:
            flat = ch.reshape(-1, 2)
            for p in np.unique(flat[:, 0]):
                uses[int(p)] += np.bincount(flat[flat[:, 0] == p, 1], minlength=layers[int(p)].num_folds)
    for p, f in out_pairs:
        uses[int(p)][int(f)] += 1
    regions: list[RegionBlock] = []
    absorbed: dict[int, np.ndarray] = {}
    for j, (s, l) in enumerate(zip(plan.layers, layers)):
        if j in skip or s.type != "sum" or not getattr(l, "_mixing", False) or l.num_output_units not in CP_K:
            continue
        ch = children[j]  # (F, H, 2)
        prods = [int(p) for p in np.unique(ch[..., 0])]
        # a bare Hadamard layer (no dense fold to absorb) counts as a block of plain slots: the region
        # launch then computes mixing-of-products straight from the Hadamard's own inputs
        bare = {p: CPBlock(p, children[p].copy(), np.full_like(children[p], -1)) for p in prods
                if (p not in cp_blocks and p not in skip and plan.layers[p].type == "hadamard" and not layers[p].is_complex
                    and layers[p].num_input_units == l.num_output_units)}
        blk = {**cp_blocks, **bare}
        if any(p not in blk or blk[p].post for p in prods):
            continue
        arities = {blk[p].slot_child.shape[1] for p in prods}
        if len(arities) != 1:
            continue
        if any((uses[p][ch[..., 1][ch[..., 0] == p]] != 1).any() for p in prods):
            continue
        cp_blocks.update(bare)
        S = arities.pop()
        F, H = ch.shape[:2]
        slot_child = np.zeros((F, H, S, 2), dtype=np.int64)
        slot_dense = np.zeros((F, H, S, 2), dtype=np.int64)
        for p in prods:
            sel = ch[..., 0] == p
            folds = ch[..., 1][sel]
            slot_child[sel] = cp_blocks[p].slot_child[folds]
            slot_dense[sel] = cp_blocks[p].slot_dense[folds]
            absorbed.setdefault(p, np.zeros(layers[p].num_folds, dtype=bool))[folds] = True
        regions.append(RegionBlock(j, slot_child, slot_dense))
    return regions, absorbed


# ---------------------------------------------------------------------------------------------
# dense layers over a Categorical layer: one table row per category instead of one MFMA per batch row
# ---------------------------------------------------------------------------------------------
def find_table_dense(plan, layers, children, skip: set[int]) -> dict[int, int]:
    """{dense layer: categorical layer} for dense layers (arity-1 sums, 32 -> 32 or 64 -> 64 units, weights = plain
    softmax) ALL of whose inputs are folds of one Categorical layer with probs = plain softmax.  Such a
    layer takes C distinct values per fold: T'[d] = dense_d(log-table of its leaf fold) is built once per
    forward by the prologue (ck_param.hip, kind 4) and the layer -- or the CP-block slot that absorbed it
    -- gathers rows of T' by the batch values (the `dense_on_table` idea beyond the leaf subtree)."""
    if plan.semiring != "lse-sum":
        return {}
    found: dict[int, int] = {}
    for j, (s, l) in enumerate(zip(plan.layers, layers)):
        if (j in skip or s.type != "sum" or l.arity != 1 or getattr(l, "_mixing", False) or l.is_complex
                or l.num_input_units not in CP_K or l.num_output_units != l.num_input_units):
            continue
        ch = children[j]
        prods = np.unique(ch[..., 0])
        if len(prods) != 1:
            continue
        c = int(prods[0])
        sc = plan.layers[c]
        if c in skip or sc.type != "categorical" or sc.scope_idx.shape[1] != 1:
            continue
        found[j] = c
    return found


def leaf_segments(num_roots: int, num_tiles: int, num_wg: int) -> np.ndarray:
    """Segment list ``(n_seg, 4)`` int32 -- rows ``{root fold, first tile, end tile, 0}`` -- of the persistent leaf
    launch (`ck_leaf_persistent_fwd`): workgroup g of `num_wg` takes segments g, g + num_wg, ...

    Segments never straddle two roots (a root's 2^D - 1 weight matrices are staged in LDS once per segment).  With
    at least as many workgroups as roots every workgroup gets ONE segment: root r owns the slots
    ``{p : p * num_roots // num_wg == r}`` (floor or ceil of num_wg / num_roots of them) and its tiles are split
    evenly among them; slot p is given to workgroup ``b`` with ``p = (b % 8) * (num_wg / 8) + b // 8`` -- workgroup b
    runs on XCD b % 8 (MI355X_MICROARCH.md), so the workgroups sharing a root's table rows and weights share one L2.
    With more roots than workgroups the segments are whole roots dealt round-robin.
    """
    if num_roots >= num_wg:
        seg = [[r, 0, num_tiles, 0] for r in range(num_roots)]
        return np.asarray(seg, dtype=np.int32).reshape(-1, 4)
    slots = [[] for _ in range(num_roots)]
    for p in range(num_wg):
        slots[p * num_roots // num_wg].append(p)
    by_slot = np.zeros((num_wg, 4), dtype=np.int32)
    for r, ps in enumerate(slots):
        c = len(ps)
        for j, p in enumerate(ps):
            by_slot[p] = (r, j * num_tiles // c, (j + 1) * num_tiles // c, 0)
    if num_wg % 8 == 0:
        b = np.arange(num_wg)
        return np.ascontiguousarray(by_slot[(b % 8) * (num_wg // 8) + b // 8])
    return by_slot


def balanced_segments(num_roots: int, num_tiles: int, num_wg: int, waves: int = 8, max_pieces: int = 16,
                      overhead: float = 1.5, xcd_aware: bool = True) -> np.ndarray:
    """Segment list ``(n_seg, 4)`` for a persistent launch whose workgroups take SEVERAL segments (`ck_leaf_walk_bwd`:
    workgroup g takes segments g, g + num_wg, ...).  A segment's tiles are dealt round-robin to the `waves` waves of its
    workgroup, so it costs ceil(tiles / waves) unit times plus `overhead` (weights staged, weight gradients flushed: measured ~1.5 unit times, scripts/bwd_stamps.py), and
    the launch takes as long as its busiest workgroup.  Every root is therefore cut into k ranges of WHOLE wave rounds
    (multiples of `waves` tiles, the remainder in the last one), k chosen to minimise that critical path, and the segments are
    dealt longest first.  196 roots x 128 tiles on 256 workgroups of 8 waves: one segment per root leaves 60 workgroups
    idle and 16 units per wave on the others (ideal: 12.25); k = 5 (3, 3, 3, 3, 4 rounds) gets 14."""
    rounds = -(-num_tiles // waves)  # wave rounds per root

    def pieces(k: int) -> list[tuple[int, int]]:
        out, r0 = [], 0
        for j in range(k):
            r1 = (j + 1) * rounds // k
            if r1 > r0:
                out.append((r0 * waves, min(r1 * waves, num_tiles)))
            r0 = r1
        return out

    def schedule(k: int):
        segs = [(b - a, r, a, b) for r in range(num_roots) for a, b in pieces(k)]
        segs.sort(key=lambda t: (-t[0], t[1], t[2]))  # longest first; a root's equal pieces stay adjacent
        cost = np.zeros(num_wg)
        for i, (n, _, _, _) in enumerate(segs):
            cost[i % num_wg] += -(-n // waves) + overhead
        return float(cost.max()), segs

    best = min((schedule(k) for k in range(1, max(1, min(max_pieces, rounds)) + 1)), key=lambda t: t[0])
    # ... or the flat list of (root, wave round) units cut into num_wg contiguous, equally long stretches: a stretch crosses at
    # most one root boundary when a root has at least as many rounds as a stretch, i.e. a workgroup takes at most two segments
    # (its second one may be empty: first tile = end tile).  196 roots x 16 rounds on 256 workgroups of 8 waves: 13 rounds + two
    # overheads = 16 against 17.5 for whole roots on 196 of the 256 workgroups.
    total = num_roots * rounds
    if total >= num_wg and rounds * num_wg >= total:
        first = np.zeros((num_wg, 4), dtype=np.int32)
        second = np.zeros((num_wg, 4), dtype=np.int32)
        worst = 0.0
        for g in range(num_wg):
            u0, u1 = g * total // num_wg, (g + 1) * total // num_wg
            r0, r1 = u0 // rounds, (u1 - 1) // rounds
            a0 = (u0 - r0 * rounds) * waves
            if r0 == r1:
                first[g] = (r0, a0, min((u1 - r0 * rounds) * waves, num_tiles), 0)
                second[g] = (r0, 0, 0, 0)
                cost = (u1 - u0) + overhead
            else:  # (r1 == r0 + 1: a stretch is no longer than a root)
                first[g] = (r0, a0, num_tiles, 0)
                second[g] = (r1, 0, min((u1 - r1 * rounds) * waves, num_tiles), 0)
                cost = (u1 - u0) + 2 * overhead
            worst = max(worst, cost)
        if worst < best[0]:
            return np.ascontiguousarray(np.concatenate([first, second]))  # (neighbouring workgroups share roots: dealt as they are)
    segs = np.asarray([[r, a, b, 0] for _, r, a, b in best[1]], dtype=np.int32).reshape(-1, 4)
    if xcd_aware and num_wg % 8 == 0 and len(segs) >= num_wg:
        # workgroup g runs on XCD g % 8: within every dealing round hand position s to workgroup 8 (s % (num_wg / 8)) +
        # s // (num_wg / 8), so that neighbouring positions -- the pieces of ONE root, which gather from the same tables and
        # stage the same weights -- run side by side on one XCD (one L2) instead of on eight
        per = num_wg // 8
        g = np.arange(num_wg)
        perm = (g % 8) * per + g // 8  # position whose segment workgroup g takes
        out = segs.copy()
        full = len(segs) // num_wg * num_wg
        for r0 in range(0, full, num_wg):  # (a last partial round stays as dealt)
            out[r0:r0 + num_wg] = segs[r0 + perm]
        return out
    return segs


def tensordot_lists(layers, children, out_layers: set[int], busy: set[int] = frozenset()) -> tuple[dict[int, int], dict[int, int]]:
    """What the TensorDot launches of a squared circuit's partition function take over (`ck_tensordot_lse_fwd_h`,
    `ck_tensordot2_lse_fwd / _bwd`, cirkit_amd/csrc/ck_backward_c.hip; TorchTensorDotLayer, optimized.py:287-300, over
    TorchHadamardLayer, inner.py:126-127):

    * ``had_of[a] = h``: the Hadamard layer h whose folds TensorDot layer a reads one to one and nobody else reads -- a reads h's
      children as a list, h is never launched;
    * ``pair_of[b] = a``: TensorDot layer b over TensorDot layer a, fold by fold, a read by nobody else (the W and conj W halves
      of a squared sum layer, M' = W M W^T): one launch for both.

    `out_layers`: layers that hold a circuit output (never absorbed); `busy`: layers another fusion already owns."""
    from .layers import HipHadamardLayer, HipTensorDotLayer

    readers: dict[int, set[int]] = {}
    for j, ch in enumerate(children):
        if ch is not None:
            for p in np.unique(ch[..., 0]):
                readers.setdefault(int(p), set()).add(j)

    def one_to_one(j: int):
        ch = children[j]
        if ch is None or ch.shape[1] != 1 or len(np.unique(ch[..., 0])) != 1:
            return None
        p = int(ch[0, 0, 0])
        if layers[p].num_folds != layers[j].num_folds or not np.array_equal(ch[:, 0, 1], np.arange(layers[j].num_folds)):
            return None
        return p if readers.get(p) == {j} and p not in out_layers and p not in busy else None

    had_of: dict[int, int] = {}
    pair_of: dict[int, int] = {}
    for j, l in enumerate(layers):
        if not isinstance(l, HipTensorDotLayer) or j in busy:
            continue
        p = one_to_one(j)
        if p is None:
            continue
        lp = layers[p]
        if isinstance(lp, HipHadamardLayer):
            had_of[j] = p
        elif (isinstance(lp, HipTensorDotLayer) and p not in pair_of and l._num_contract_units == lp._num_batch_units
              and l._num_batch_units == lp.num_output_units // lp._num_batch_units):
            pair_of[j] = p
    return had_of, pair_of
