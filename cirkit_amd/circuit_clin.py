"""Complex-valued circuits on linear (re, im) tiles (`csrc/ck_clin.hip`): the launch list `HipCircuit` takes for a
complex-lse-sum circuit of 32-unit CP-T / dense layers over ONE Embedding layer whose parameters are plain tensors, real or
complex (BASELINE config 5 with complex-valued parameters; real parameters take the signed tiles of ck_leaf.hip instead).

Replaces, for such a circuit, `TorchEmbeddingLayer.forward` (layers/input.py:258-266), `TorchCPTLayer.forward` /
`TorchSumLayer.forward` (layers/optimized.py:171-178, inner.py:266-273) under `ComplexLSESumSemiring.apply_reduce`
(semiring.py:441-476): values travel between launches as (re + i im) 2^e tiles, the complex logarithm is taken once, by the
layer the circuit outputs.  Launches per forward: table, leaf (Embedding + `depth` CP-T levels), one per remaining layer.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import TYPE_CHECKING

import numpy as np
import torch

from . import _capi as capi
from .fusion import find_subtree_groups
from .layers import HipEmbeddingLayer

if TYPE_CHECKING:
    from .circuit import HipCircuit, _Binding

_TILE = 2048  # floats of one (fold, 32-row tile): 1024 re + 1024 im
# ck_clin_tail_fold of include/cirkit_hip.h
_TAIL_FOLD = np.dtype([("co", "<i8", (2,)), ("ce", "<i8", (2,)), ("w", "<u8"), ("out", "<i8"), ("oute", "<i8"), ("out_log", "<u8"),
                       ("H", "<i4"), ("Ko", "<i4")])
assert _TAIL_FOLD.itemsize == 72
_TAIL_MAX_FOLDS = 64


def _plain_tensor(param) -> torch.Tensor | None:
    """The stored tensor behind a parameter whose graph is ONE tensor node read with the identity fold index (what
    TorchParameter.forward returns without any kernel, parameters/parameter.py:180-188), else None."""
    from .plan import IDX_NONE

    g = param.graph
    if g.ops != ["tensor"]:
        return None
    n = g.nodes[0]
    if g.output.ids != [0] or not (g.output.kind == IDX_NONE or np.array_equal(
            np.asarray(g.output.array).reshape(-1), np.arange(n.num_folds))):
        return None
    return param.store[n.config["tensor"]]


def _steps_after(i: int) -> int:
    n = 0
    while i & 1:
        n += 1
        i >>= 1
    return n


class ClinPath:
    """What `HipCircuit` needs to evaluate a qualifying complex circuit on linear tiles."""

    @staticmethod
    def build(c: "HipCircuit", depth: int) -> "ClinPath | None":
        plan = c.plan
        if not c._complex or plan.num_variables == 0 or os.environ.get("CK_CLIN", "1") == "0":
            return None
        emb = [i for i in range(len(c.layers)) if plan.layers[i].inputs is None]
        if len(emb) != 1 or not isinstance(c.layers[emb[0]], HipEmbeddingLayer):
            return None
        e = emb[0]
        le = c.layers[e]
        if le.num_output_units != 32 or le.scope_idx.shape[1] != 1 or _plain_tensor(le.weight) is None or le.num_states >= 65535:
            return None
        outs = {int(p) for p in c._out_pairs[:, 0]}
        inner = [i for i in range(len(c.layers)) if i != e]
        for i in inner:
            s, l = plan.layers[i], c.layers[i]
            if not (s.type == "cpt" or (s.type == "sum" and l.arity == 1)) or getattr(l, "_mixing", False):
                return None
            if l.num_input_units != 32 or l.num_output_units > 32 or getattr(l, "weight", None) is None or _plain_tensor(l.weight) is None:
                return None
            if l.num_output_units < 32 and any(i in {int(p) for p in np.unique(ch[..., 0])} for ch in c._children if ch is not None):
                return None  # (fewer than 32 units: only the layer the circuit outputs)
            if l.num_folds > 65535:
                return None
        depth = int(os.environ.get("CK_CLIN_DEPTH", min(max(1, depth), 3)))
        groups = find_subtree_groups(plan, c.layers, c._children, c._out_pairs, depth, signed=True)
        groups = [g for g in groups if g.input_layer == e and g.depth >= 1]
        if len(groups) != 1:
            return None
        g = groups[0]
        rest = [i for i in inner if i not in g.levels]
        if not rest or outs & set(g.virtual) or g.root in outs:
            return None
        wcx = {i: bool(_plain_tensor(c.layers[i].weight).is_complex()) for i in inner}
        if len({wcx[i] for i in g.levels}) != 1:
            return None  # (one launch walks all its levels with one kind of weights)
        if any(int(p) not in rest for p in outs):
            return None
        return ClinPath(c, e, g, rest, wcx)

    def __init__(self, c: "HipCircuit", emb: int, group, rest: list[int], wcx: dict[int, bool]):
        self.c, self.emb, self.group, self.rest, self.wcx = c, emb, group, rest, wcx
        self.outs = {int(p) for p in c._out_pairs[:, 0]}
        le = c.layers[emb]
        self.C = int(le.num_states)
        self.table_complex = bool(_plain_tensor(le.weight).is_complex())
        dev = c.device
        F0 = le.num_folds
        self.table = torch.empty((F0, self.C + 1, 64 if self.table_complex else 32), dtype=torch.float32, device=dev)
        self.table_e = torch.empty((F0, self.C + 1), dtype=torch.int32, device=dev)
        # the leaf launch's tables: leaves in walk order, weights in contraction order
        g = group
        D = g.depth
        R = c.layers[g.root].num_folds
        nodes = np.asarray(g.nodes, dtype=np.int64)
        tabs = [nodes[g.node_off[lv]: (g.node_off[lv + 1] if lv < D else g.leaf_off)].reshape(R, -1) for lv in range(D + 1)]
        leaf = nodes[g.leaf_off:].reshape(R, 1 << D)
        self.leaf_fold = torch.from_numpy(leaf.astype(np.int32)).to(dev)
        self.leaf_var = torch.from_numpy(np.asarray(le.scope_idx)[leaf, 0].astype(np.int32)).to(dev)
        self._node_fold: list[tuple[int, np.ndarray]] = []  # (layer, fold per root) per contraction of the walk
        for i in range(1 << D):
            for l in range(_steps_after(i)):
                self._node_fold.append((g.levels[l], tabs[l + 1][:, i >> (l + 1)]))
        self.R, self.D = R, D
        self.raw_batch = os.environ.get("CK_CLIN_RAW", "0") == "1"  # (lab switch: the leaf launch reads the caller's int64 batch)
        # the few-fold top of the circuit: ONE launch (ck_clin_tail_fwd) instead of one per layer -- the trailing layers with at
        # most 64 folds in all, one kind of weights, at most two children per fold
        self.tail: list[int] = []
        if os.environ.get("CK_CLIN_TAIL", "1") != "0":
            n = 0
            for i in reversed(rest):
                l = c.layers[i]
                if l.arity > 2 or n + l.num_folds > _TAIL_MAX_FOLDS or (self.tail and wcx[i] != wcx[self.tail[0]]):
                    break
                self.tail.insert(0, i)
                n += l.num_folds
            if len(self.tail) < 2:
                self.tail = []
        self._wnode: torch.Tensor | None = None
        self._wlayer: dict[int, torch.Tensor] = {}
        self._wkey = None

    # ------------------------------------------------------------------------------------------------------------------
    def virtual_layers(self) -> set[int]:
        """Layers that never reach the activation arena: everything but the layers the circuit outputs."""
        return {i for i in range(len(self.c.layers)) if i not in self.outs}

    def _weight_tables(self) -> None:
        """Device tables of weight-matrix addresses (rebuilt when the store replaced a tensor)."""
        c = self.c
        key = tuple(int(_plain_tensor(c.layers[i].weight).data_ptr()) for i in sorted(self.wcx))
        if key == self._wkey:
            return
        self._wkey = key

        def base(i: int) -> tuple[int, int]:
            w = _plain_tensor(c.layers[i].weight)
            return int(w.data_ptr()), int(w.shape[-2]) * int(w.shape[-1]) * (8 if w.is_complex() else 4)

        cols = []
        for layer, folds in self._node_fold:
            b, per = base(layer)
            cols.append(b + folds.astype(np.int64) * per)
        self._wnode = torch.from_numpy(np.ascontiguousarray(np.stack(cols, axis=1))).to(c.device)  # (R, 2^D - 1)
        for i in self.rest:
            b, per = base(i)
            self._wlayer[i] = torch.from_numpy(b + np.arange(c.layers[i].num_folds, dtype=np.int64) * per).to(c.device)

    def bind(self, bd: "_Binding") -> None:
        """The tile blocks of a batch size: the leaf group's root and every remaining layer that somebody reads."""
        c, B = self.c, bd.B
        tiles = (B + 31) // 32
        mat = [self.group.root] + [i for i in self.rest if i not in self.outs or self._is_read(i)]
        base, ebase, off, eoff = {}, {}, 0, 0
        for i in mat:
            base[i], ebase[i] = off, eoff
            off += c.layers[i].num_folds * tiles * _TILE
            eoff += c.layers[i].num_folds * tiles * 32
        bd.clin = {
            "lin": torch.empty(max(off, 1), dtype=torch.float32, device=c.device),
            "lin_e": torch.empty(max(eoff, 1), dtype=torch.int32, device=c.device),
            "base": base, "ebase": ebase, "tiles": tiles, "child": {},
        }
        for i in self.rest:
            ch = c._children[i]
            prod, fold = ch[..., 0], ch[..., 1]
            co = np.vectorize(base.__getitem__)(prod).astype(np.int64) + fold.astype(np.int64) * (tiles * _TILE)
            ce = np.vectorize(ebase.__getitem__)(prod).astype(np.int64) + fold.astype(np.int64) * (tiles * 32)
            bd.clin["child"][i] = (torch.from_numpy(np.ascontiguousarray(co)).to(c.device),
                                   torch.from_numpy(np.ascontiguousarray(ce)).to(c.device))
            bd.clin.setdefault("child_np", {})[i] = (co, ce)

    def _is_read(self, i: int) -> bool:
        return any(ch is not None and i in {int(p) for p in np.unique(ch[..., 0])} for ch in self.c._children)

    # ------------------------------------------------------------------------------------------------------------------
    def launches(self, bd: "_Binding") -> list[tuple[int, str, object]]:
        """(layer, kernel name, fn(stream)) of every launch of one forward, in order."""
        c = self.c
        self._weight_tables()
        st = bd.clin
        le = c.layers[self.emb]
        g = self.group
        lin, lin_e = st["lin"], st["lin_e"]
        tc = 1 if self.table_complex else 0

        def table(stream):
            capi.call("ck_clin_table", _plain_tensor(le.weight).data_ptr(), tc, self.table.data_ptr(), self.table_e.data_ptr(), le.num_folds, self.C, stream)

        # the batch: the caller's int64 tensor itself (program input cell 0 while recording; validated by the launch: a bad row
        # becomes NaN and raises the circuit's flag), or the staged (D, B) int32 copy
        if bd.direct:
            x_rows, x_input = c._raw_batch_args(bd)
            xt, flag = None, (c._bad_input.data_ptr() if c.validate_inputs else None)
        else:
            xt, x_rows, x_input, flag = bd.xt_i.data_ptr(), None, -1, None

        def leaf(stream):
            capi.call("ck_clin_leaf_fwd", self.table.data_ptr(), self.table_e.data_ptr(), xt, x_rows, x_input, c.plan.num_variables, flag,
                      self.leaf_fold.data_ptr(), self.leaf_var.data_ptr(), self._wnode.data_ptr(), 1 if self.wcx[g.levels[0]] else 0, tc,
                      lin.data_ptr() + 4 * st["base"][g.root], lin_e.data_ptr() + 4 * st["ebase"][g.root], self.R, self.D, bd.B, self.C,
                      stream)

        def layer(i):
            l = c.layers[i]
            co, ce = st["child"][i]
            out = lin.data_ptr() + 4 * st["base"][i] if i in st["base"] else None
            out_e = lin_e.data_ptr() + 4 * st["ebase"][i] if i in st["ebase"] else None
            out_log = bd.views[i].data_ptr() if i in self.outs else None

            def fn(stream):
                capi.call("ck_clin_layer_fwd", lin.data_ptr(), lin_e.data_ptr(), co.data_ptr(), ce.data_ptr(), self._wlayer[i].data_ptr(),
                          1 if self.wcx[i] else 0, out, out_e, out_log, l.num_folds, l.arity, l.num_output_units, bd.B, stream)
            return fn

        wk = "WCplx" if self.wcx[g.levels[0]] else "WReal"
        rows = [(self.emb, "clin_table_kernel", table),
                (g.root, f"clin_leaf_kernel<{self.D}, {wk}, {'true' if self.table_complex else 'false'}>", leaf)]
        rows += [(i, f"clin_layer_kernel<{'WCplx' if self.wcx[i] else 'WReal'}>", layer(i)) for i in self.rest if i not in self.tail]
        if self.tail:
            desc, level_off = self._tail_tables(bd)
            tw = 1 if self.wcx[self.tail[0]] else 0

            def tail(stream):
                capi.call("ck_clin_tail_fwd", lin.data_ptr(), lin_e.data_ptr(), desc.data_ptr(), level_off.data_ptr(), len(self.tail), tw, bd.B, stream)

            rows.append((self.tail[-1], f"clin_tail_kernel<{'WCplx' if tw else 'WReal'}>", tail))
        return rows

    def _tail_tables(self, bd: "_Binding") -> tuple[torch.Tensor, torch.Tensor]:
        """The fold descriptors of the tail launch for this batch size (rebuilt when a weight tensor was replaced)."""
        c, st = self.c, bd.clin
        if st.get("tail_key") == self._wkey:
            return st["tail_desc"], st["tail_levels"]
        tiles, B = st["tiles"], bd.B
        n = sum(c.layers[i].num_folds for i in self.tail)
        d = np.zeros(n, dtype=_TAIL_FOLD)
        level_off, k = [0], 0
        for i in self.tail:
            l = c.layers[i]
            w = _plain_tensor(l.weight)
            per = int(w.shape[-2]) * int(w.shape[-1]) * (8 if w.is_complex() else 4)
            co, ce = st["child_np"][i]
            for f in range(l.num_folds):
                d["co"][k, : l.arity] = co[f]
                d["ce"][k, : l.arity] = ce[f]
                d["w"][k] = int(w.data_ptr()) + f * per
                d["out"][k] = st["base"][i] + f * tiles * _TILE if i in st["base"] else -1
                d["oute"][k] = st["ebase"][i] + f * tiles * 32 if i in st["ebase"] else -1
                d["out_log"][k] = (bd.views[i].data_ptr() + f * B * l.num_output_units * 8) if i in self.outs else 0
                d["H"][k], d["Ko"][k] = l.arity, l.num_output_units
                k += 1
            level_off.append(k)
        st["tail_desc"] = torch.from_numpy(d.view(np.uint8).reshape(-1)).to(c.device)
        st["tail_levels"] = torch.tensor(level_off, dtype=torch.int32, device=c.device)
        st["tail_key"] = self._wkey
        return st["tail_desc"], st["tail_levels"]

    def enqueue(self, bd: "_Binding", stream: int) -> None:
        for _, _, fn in self.launches(bd):
            fn(stream)

    def profile(self, bd: "_Binding", iters: int) -> list[dict]:
        """HIP events around every launch of `iters` eager forwards on the current stream (the batch staged by the caller)."""
        c = self.c
        cur = torch.cuda.current_stream(c.device)
        ls = self.launches(bd)
        tot = [0.0] * len(ls)
        for it in range(iters + 1):
            evs = []
            for _, _, fn in ls:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(cur)
                fn(cur.cuda_stream)
                b.record(cur)
                evs.append((a, b))
            torch.cuda.synchronize(c.device)
            if it:
                for k, (a, b) in enumerate(evs):
                    tot[k] += a.elapsed_time(b)
        rows = []
        for k, (i, name, _) in enumerate(ls):
            l = c.layers[i]
            n_contr = 4 if (i != self.emb and self.wcx.get(i, False)) else 2  # real 32 x 32 contractions per complex one
            if i == self.group.root:
                folds = sum(c.layers[j].num_folds for j in self.group.levels)
            elif i == self.emb:
                folds = 0
            elif name.startswith("clin_tail_kernel"):
                folds = sum(c.layers[j].num_folds for j in self.tail)
            else:
                folds = l.num_folds
            fl = folds * bd.B * 2.0 * 32 * 32 * n_contr
            rows.append({"layer": i, "kernel": name, "ms": tot[k] / max(iters, 1), "algorithmic_bytes": 0.0, "algorithmic_flops": fl,
                         "executed_flops": fl})
        return rows
