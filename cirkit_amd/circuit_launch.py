"""The launches of a `HipCircuit` forward, one method per kind (split out of circuit.py; `HipCircuit` inherits this mixin).

What each launch of the recorded program needs -- device tables of weight addresses and offsets, fold descriptors of the
tail walk, the table jobs of the parameter launch, the leaf walk's node tables -- and the C-ABI call itself: TensorDot
pairs, Embedding gathers, tabulated dense layers, regions / CP blocks (K = 64), the tail (+ parameters of the next forward),
leaf groups.  Replaces the per-layer bodies of `TorchDiAcyclicGraph.evaluate` (graph/modules.py:303-335) for fused launches.
"""

from __future__ import annotations

import ctypes as C
from typing import TYPE_CHECKING

import numpy as np
import torch

from . import _capi as capi
from .fusion import SubtreeGroup, leaf_segments
from .parameters import ParamBatch

if TYPE_CHECKING:
    from .circuit import _Binding

# ck_tail16_fold of include/cirkit_hip.h
_TAIL16_FOLD = np.dtype([("w", "<u8"), ("out", "<u8"), ("child", "<u8", (4,)), ("child_src", "<i4", (4,)), ("H", "<i4"),
                         ("Ko", "<i4"), ("skip_store", "<i4"), ("slot", "<i4")])
assert _TAIL16_FOLD.itemsize == 80


class _LaunchMixin:
    def _launch_tensordot(self, i: int, bd: _Binding, stream: int) -> None:
        """TensorDot layer i with what it absorbed (`_td_had`, `_td_pair`): `ck_tensordot_lse_fwd_h` / `ck_tensordot2_lse_fwd`."""
        l = self.layers[i]
        a = self._td_pair.get(i)
        first = i if a is None else a
        h = self._td_had.get(first)
        ro, H = (bd.row_off[first], 1) if h is None else (bd.row_off[h], self.layers[h].arity)
        cv = 1 if self._complex else 0
        if a is None:
            capi.call("ck_tensordot_lse_fwd_h", bd.arena.data_ptr(), ro.data_ptr(), H, l._w.data_ptr(), bd.views[i].data_ptr(), l.num_folds,
                      bd.B, l._num_contract_units, l._num_batch_units, l.num_output_units // l._num_batch_units, cv, stream)
        else:
            la = self.layers[a]
            capi.call("ck_tensordot2_lse_fwd", bd.arena.data_ptr(), ro.data_ptr(), H, la._w.data_ptr(), bd.views[a].data_ptr(),
                      l._w.data_ptr(), bd.views[i].data_ptr(), l.num_folds, bd.B, la._num_contract_units, la._num_batch_units,
                      la.num_output_units // la._num_batch_units, l.num_output_units // l._num_batch_units, cv, stream)

    def _launch_emb_gather(self, i: int, bd: _Binding, stream: int) -> None:
        """`ck_sum_clse_gather_fwd`: a complex CP-T / dense layer reading its Embedding children from the table."""
        l, emb = self.layers[i], self.layers[self._emb_gather[i]]
        tabs = self._emb_gather_dev.get(i)
        if tabs is None:
            folds = self._children[i][..., 1].astype(np.int32)
            variables = emb.scope_idx[folds, 0].astype(np.int32)
            tabs = self._emb_gather_dev[i] = (torch.from_numpy(np.ascontiguousarray(folds)).to(self.device),
                                              torch.from_numpy(np.ascontiguousarray(variables)).to(self.device))
        if l._w.is_complex():
            raise ValueError("gathering CP-T layers take real weights")
        capi.call("ck_sum_clse_gather_fwd", emb._table.data_ptr(), bd.xt_i.data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr(),
                  l._w.data_ptr(), bd.views[i].data_ptr(), l.num_folds, l.arity, bd.B, emb.num_states, stream)

    def _launch_table_dense(self, i: int, bd: _Binding, stream: int) -> None:
        """A dense layer over a Categorical layer, evaluated as a gather from its per-category table T'
        (all folds, or only those a consumer outside the CP blocks still reads)."""
        l = self.layers[i]
        table, scope, _ = self._tdense_dev[i]
        Cn = self.layers[self._tdense[i]].num_categories
        sub = self._cp_leftover.get(i)
        if sub is None:
            capi.call("ck_categorical_fwd", table.data_ptr(), bd.xt_i.data_ptr(), scope.data_ptr(), bd.views[i].data_ptr(),
                      l.num_folds, bd.B, l.num_output_units, Cn, self.plan.num_variables, stream)
            return
        for f in sub:  # a handful of folds
            capi.call("ck_categorical_fwd", table[int(f)].data_ptr(), bd.xt_i.data_ptr(), scope[int(f) : int(f) + 1].data_ptr(),
                      bd.views[i][int(f)].data_ptr(), 1, bd.B, l.num_output_units, Cn, self.plan.num_variables, stream)

    def _gather_tables(self, slot_dense: np.ndarray, key, bd: _Binding):
        """(g_addr, g_var, C) device tables for the slots of `slot_dense` whose dense layer is tabulated, or Nones."""
        if not any(int(d) in self._tdense for d in np.unique(slot_dense[..., 0]) if d >= 0):
            return None, None, 0
        tabs = bd.cp_tabs.get((key, "gather"))
        Cn = 0
        if tabs is None:
            addr = np.zeros(slot_dense.shape[:-1], dtype=np.int64)
            var = np.full(slot_dense.shape[:-1], -1, dtype=np.int32)
            for d in np.unique(slot_dense[..., 0]):
                if d < 0 or int(d) not in self._tdense:
                    continue
                table, _, variables = self._tdense_dev[int(d)]
                sel = slot_dense[..., 0] == d
                folds = slot_dense[..., 1][sel]
                addr[sel] = table.data_ptr() + folds * (table.shape[1] * table.shape[2] * 4)
                var[sel] = variables[folds]
            tabs = bd.cp_tabs[(key, "gather")] = (torch.from_numpy(addr).to(self.device), torch.from_numpy(var).to(self.device))
        cats = {self.layers[self._tdense[int(d)]].num_categories for d in np.unique(slot_dense[..., 0]) if int(d) in self._tdense}
        if len(cats) != 1:  # (cannot happen: __init__ tabulates one category count only)
            raise ValueError(f"gather slots over tables with different numbers of categories {sorted(cats)}")
        return tabs[0], tabs[1], cats.pop()

    def _weight_addresses(self, slot_dense: np.ndarray, K: int) -> np.ndarray:
        """Device addresses of the (K, K) weight matrices of the dense folds in `slot_dense` (0 = none)."""
        addr = np.zeros(slot_dense.shape[:-1], dtype=np.int64)
        for d in np.unique(slot_dense[..., 0]):
            if d < 0 or int(d) in self._tdense:  # tabulated dense layers are gather slots without weights
                continue
            w = self.layers[int(d)]._w
            if w.is_complex() or w.dtype != torch.float32 or not w.is_contiguous():
                raise ValueError("CP blocks need real, contiguous fp32 weights")
            sel = slot_dense[..., 0] == d
            addr[sel] = w.data_ptr() + slot_dense[..., 1][sel] * (K * K * 4)
        return addr

    def _launch_region(self, i: int, bd: _Binding, stream: int) -> None:
        """`ck_region_lse_fwd`: a mixing layer together with the CP blocks it combines."""
        l = self.layers[i]
        reg = self._regions[i]
        K = l.num_output_units
        tab = bd.cp_tabs.get(i)
        if tab is None:
            tab = bd.cp_tabs[i] = torch.from_numpy(self._weight_addresses(reg.slot_dense, K)).to(self.device)
        F, H, S = reg.slot_dense.shape[:3]
        ga, gv, Cn = self._gather_tables(reg.slot_dense, i, bd)
        redo = None
        if self.linear_levels and ga is None:  # linear-space products + marked workgroups again in log space
            redo = bd.cp_tabs.get((i, "redo"))
            if redo is None:
                redo = bd.cp_tabs[(i, "redo")] = torch.zeros(F * ((bd.B + 127) // 128), dtype=torch.int32, device=self.device)
        capi.call("ck_region_lse_fwd_v", bd.arena.data_ptr(), bd.row_off[i].data_ptr(), tab.data_ptr(), l._w.data_ptr(),
                  bd.views[i].data_ptr(), None if ga is None else ga.data_ptr(), None if gv is None else gv.data_ptr(),
                  None if ga is None else bd.xt_i.data_ptr(), Cn, None if redo is None else redo.data_ptr(),
                  F, H, S, bd.B, K, self._ct, stream)

    def _launch_input_prod(self, i: int, bd: _Binding, stream: int) -> None:
        """`ck_gaussian_prod_fwd`: a Hadamard layer over Gaussian folds, straight from the batch."""
        g = self.layers[self._input_prod[i]]
        tab = self._input_prod_dev.get(i)
        if tab is None:
            tab = self._input_prod_dev[i] = torch.from_numpy(
                np.ascontiguousarray(self._children[i][..., 1].astype(np.int32))).to(self.device)
        mean, stddev, lz = g._vals
        l = self.layers[i]
        capi.call("ck_gaussian_prod_fwd", mean.data_ptr(), stddev.data_ptr(), None if lz is None else lz.data_ptr(),
                  bd.xt.data_ptr(), g._scope(self.device).data_ptr(), tab.data_ptr(), bd.views[i].data_ptr(),
                  l.num_folds, l.arity, bd.B, l.num_output_units, stream)

    def _launch_cp(self, i: int, bd: _Binding, stream: int) -> None:
        """`ck_cp_lse_fwd`: a Hadamard layer with its dense layers folded in, or the folds of a dense
        layer that consumers outside such blocks still read."""
        l = self.layers[i]
        K = l.num_output_units
        tab = bd.cp_tabs.get(i)
        if i in self._cp_blocks:
            blk = self._cp_blocks[i]
            sub = self._cp_subset.get(i)
            if tab is None:
                addr = self._weight_addresses(blk.slot_dense, K)
                tab = bd.cp_tabs[i] = torch.from_numpy(np.ascontiguousarray(addr if sub is None else addr[sub])).to(self.device)
            F, S = blk.slot_dense.shape[:2]
            post = None
            if blk.post:  # the CP-T layer's own weights, one matrix per evaluated fold
                post = bd.cp_tabs.get((i, "post"))
                if post is None:
                    folds = np.arange(F, dtype=np.int64) if sub is None else sub.astype(np.int64)
                    if l._w.is_complex() or not l._w.is_contiguous():
                        raise ValueError("CP blocks need real, contiguous fp32 weights")
                    post = bd.cp_tabs[(i, "post")] = torch.from_numpy(l._w.data_ptr() + folds * (K * K * 4)).to(self.device)
            pp = None if post is None else post.data_ptr()
            ga, gv, Cn = self._gather_tables(blk.slot_dense if sub is None else blk.slot_dense[sub], i, bd)
            gargs = (None if ga is None else ga.data_ptr(), None if gv is None else gv.data_ptr(),
                     None if ga is None else bd.xt_i.data_ptr(), Cn)
            if sub is None:
                capi.call("ck_cp_lse_fwd_v", bd.arena.data_ptr(), bd.row_off[i].data_ptr(), tab.data_ptr(), pp, None,
                          bd.views[i].data_ptr(), *gargs, F, S, 1, bd.B, K, self._ct, stream)
            else:
                ro, oo = bd.leftover[i]
                capi.call("ck_cp_lse_fwd_v", bd.arena.data_ptr(), ro.data_ptr(), tab.data_ptr(), pp, oo.data_ptr(),
                          bd.arena.data_ptr(), *gargs, len(sub), S, 1, bd.B, K, self._ct, stream)
            return
        folds = self._cp_leftover[i]
        ro, oo = bd.leftover[i]
        if tab is None:
            tab = bd.cp_tabs[i] = torch.from_numpy(l._w.data_ptr() + folds.astype(np.int64) * (K * K * 4)).to(self.device)
        capi.call("ck_cp_lse_fwd", bd.arena.data_ptr(), ro.data_ptr(), tab.data_ptr(), None, oo.data_ptr(),
                  bd.arena.data_ptr(), None, None, None, 0, len(folds), 1, 1, bd.B, K, stream)

    def _enqueue_params_batch_only(self, stream: int, bd: _Binding) -> None:
        """The prologue launch of a forward of this binding (profiling): all jobs, or what the leaf launch leaves."""
        if bd.params_at_end:
            self._ensure_param_batch()
            if self._tailp["rest"] is not None:
                self._tailp["rest"].launch(stream)
        else:
            self._launch_param_batch(stream)

    def _launch_param_batch(self, stream: int) -> None:
        if not self.batch_params:
            return
        self._ensure_param_batch()
        self._batch.launch(stream)

    def _ensure_param_batch(self) -> None:
        """Build the job list of the batched prologue (and with it `_table_fused`: which leaf groups read a table made by
        one of its jobs) for the current set of parameter tensors."""
        if not self.batch_params:
            return
        if self._batch is None or self._batch_version != self.store.version:
            self._batch = ParamBatch()
            self._assign_weight_layouts()
            covered = self._register_table_jobs(self._batch)
            self._jobs_of_layer: dict[int, list[int]] = {}
            for i, l in enumerate(self.layers):
                l._batched = i in covered
                if i not in covered:
                    n0 = len(self._batch._jobs)
                    l.register_batched(self._batch)
                    self._jobs_of_layer[i] = list(range(n0, len(self._batch._jobs)))
            self._tailp = self._plan_tail_params()
            self._batch_version = self.store.version

    def _plan_tail_params(self) -> dict | None:
        """Which jobs of the prologue the launch that walks the tail takes over (`params_at_end`): the table job of the
        (single) leaf group, the softmaxes of its level weights, and every other 32-wide softmax; what is left stays a
        (smaller, often empty) prologue launch.  None: nothing is taken over."""
        if not (self.params_at_end and self.batch_params and not self.cache_params
                and len(self._groups) == 1 and not self._signed):
            return None
        g = self._groups[0]
        cat = self.layers[g.input_layer]
        if (g.root not in self._table_fused or g.depth < 2 or not self.linear_levels or cat.num_categories > 256
                or cat.num_categories % 4 or self._group_layout(g) != capi.CK_W_TILED_F32):
            return None
        meta = self._batch._meta
        table = [k for k, m in enumerate(meta) if m["kind"] == 5 and m["dst"] is self._group_dev[g.root][1]]
        if len(table) != 1:
            return None
        taken = set(table)
        levels = []
        for j in g.levels:
            jobs = self._jobs_of_layer.get(j, [])
            if len(jobs) != 1 or meta[jobs[0]]["kind"] != 2 or tuple(meta[jobs[0]]["src"].shape[1:]) != (32, 32):
                return None
            levels.append(meta[jobs[0]]["src"])
            taken.add(jobs[0])
        xjobs = []
        for k, m in enumerate(meta):
            if k in taken or m["kind"] not in (0, 2) or m["src"].shape[-1] != 32 or not m["src"].is_contiguous():
                continue
            src, dst = m["src"], m["dst"]
            rows = src.numel() // 32
            per = 32 if (src.dim() >= 2 and src.shape[-2] == 32) else (rows if rows <= 32 and m["kind"] == 0 else 0)
            if per == 0 or (m["kind"] == 2 and per != 32):
                continue
            for f in range(rows // per):
                xjobs.append((src.data_ptr() + f * per * 128, dst.data_ptr() + f * per * 128, per, 1 if m["kind"] == 2 else 0))
            taken.add(k)
        job_t = np.dtype([("in", "<u8"), ("out", "<u8"), ("rows", "<i4"), ("tiled", "<i4")])
        rest = [k for k in range(len(meta)) if k not in taken]
        lv = []  # the level weights as 32-wide jobs too (for the launch that evaluates everything beside the tail)
        for j in g.levels:
            m = meta[self._jobs_of_layer[j][0]]
            for f in range(m["src"].shape[0]):
                lv.append((m["src"].data_ptr() + f * 4096, m["dst"].data_ptr() + f * 4096, 32, 1))
        xa = np.zeros(max(1, len(xjobs) + len(lv)), dtype=job_t)
        for r, t in zip(xa, xjobs + lv):
            r["in"], r["out"], r["rows"], r["tiled"] = t
        return {"root": g.root, "table": meta[table[0]],
                "rows_all": torch.from_numpy(xa.view(np.uint8)).to(self.device), "n_rows_all": len(xjobs) + len(lv),
                "rest": self._batch.subset(rest) if rest else None}

    def _params_at_end(self, B: int) -> bool:
        """Whether the tail launch of a forward at batch size B also evaluates the parameters of the next forward."""
        if not (self.params_at_end and self._tailp is not None and self._tail and self._tail16_ok() and not self._signed):
            return False
        n_slots = self._tail_slots()[2]
        return 8192 + n_slots * 2048 <= 80 * 1024 and sum(self.layers[j].num_folds for j in self._tail) * 80 + 64 <= 8192

    def _tail_slots(self) -> tuple[dict, dict, int]:
        """LDS slots for the fold tiles of the tail when LDS is tight (`ck_tail_params_fwd`): (slot of each (layer, fold),
        last level that reads it, number of slots).  A slot is reused at level L + 1 at the earliest if its fold was read
        for the last time at level L (within a level folds are read and written concurrently)."""
        hit = getattr(self, "_tail_slots_cache", None)
        if hit is not None:
            return hit
        level_of = {j: li for li, j in enumerate(self._tail)}
        last_use: dict[tuple[int, int], int] = {}
        for j in self._tail:
            for f in range(self.layers[j].num_folds):
                if self.layers[j].num_output_units == 32:
                    last_use[(j, f)] = level_of[j]
        for j in self._tail:
            ch = self._children[j]
            for f in range(ch.shape[0]):
                for h in range(ch.shape[1]):
                    key = (int(ch[f, h, 0]), int(ch[f, h, 1]))
                    if key in last_use:
                        last_use[key] = max(last_use[key], level_of[j])
        slot: dict[tuple[int, int], int] = {}
        free: list[int] = []
        busy_until: dict[int, int] = {}
        n_slots = 0
        for li, j in enumerate(self._tail):
            free += sorted(s_ for s_, u in busy_until.items() if u < li)
            for s_ in list(busy_until):
                if busy_until[s_] < li:
                    del busy_until[s_]
            for f in range(self.layers[j].num_folds):
                if (j, f) not in last_use:
                    continue
                if free:
                    sl = free.pop(0)
                else:
                    sl = n_slots
                    n_slots += 1
                slot[(j, f)] = sl
                busy_until[sl] = last_use[(j, f)]
        self._tail_slots_cache = (slot, last_use, max(1, n_slots))
        return self._tail_slots_cache

    def _register_table_jobs(self, batch: ParamBatch) -> set[int]:
        """`dense_on_table` inside the prologue: for a leaf group whose Categorical probabilities and
        dense weights are plain softmaxes, ONE job per dense fold builds the log-table and pushes it
        through the dense layer (ck_param.hip kind 4) -- neither the table nor the dense weights
        reach memory.  Returns the layers whose parameters are fully covered by such jobs."""
        covered: set[int] = set()
        self._table_fused = set()
        for d, c in self._tdense.items():  # dense layers tabulated over their Categorical layer (any plan shape)
            cat, dl = self.layers[c], self.layers[d]
            Cn = cat.num_categories
            leaf = self._children[d][:, 0, 1].astype(np.int64)
            idx = None if np.array_equal(leaf, np.arange(len(leaf))) and cat.num_folds == len(leaf) else torch.from_numpy(leaf).to(self.device)
            dst = torch.empty((dl.num_folds, Cn + 1, dl.num_output_units), dtype=torch.float32, device=self.device)
            batch.add_log_table_dense(cat.probs.softmax_source(), dl.weight.softmax_source(), idx, dst)
            variables = cat.scope_idx[leaf, 0].astype(np.int64)
            self._tdense_dev[d] = (dst, torch.from_numpy(np.ascontiguousarray(variables)).to(self.device), variables)
            covered.add(d)
            if c in self._virtual:
                covered.add(c)
        if not self.dense_on_table or (self.contraction != "f32" and not self.linear_levels):
            return covered  # (the log-space table job contracts in exact fp32 only)
        for g in self._groups:
            if g.dense_layer is None or g.depth == 0:
                continue
            cat, dl = self.layers[g.input_layer], self.layers[g.dense_layer]
            src = None if getattr(cat, "probs", None) is None else cat.probs.softmax_source()
            wsrc = dl.weight.softmax_source()
            if src is None or wsrc is None or cat.num_output_units != 32 or tuple(wsrc.shape[1:]) != (32, 32):
                continue
            Cn = cat.num_categories
            leaf = self._children[g.dense_layer][:, 0, 1].astype(np.int64)
            idx = None if np.array_equal(leaf, np.arange(len(leaf))) else torch.from_numpy(leaf).to(self.device)
            dst = torch.empty((dl.num_folds, Cn + 1, 32), dtype=torch.float32, device=self.device)
            scale = torch.empty((dl.num_folds, Cn + 1), dtype=torch.float32, device=self.device) if self.linear_levels else None
            batch.add_log_table_dense(src, wsrc, idx, dst, scale)
            dev = self._group_dev.get(g.root) or (torch.from_numpy(g.nodes).to(self.device),)
            self._group_dev[g.root] = (dev[0], dst, torch.from_numpy(np.ascontiguousarray(leaf * ((Cn + 1) * 32))).to(self.device), scale)
            self._table_fused.add(g.root)
            covered |= {g.input_layer, g.dense_layer}
        return covered

    def _tail16_ok(self) -> bool:
        """The tail as ONE launch on 16-row tiles with its fold outputs kept in LDS (ck_tail16.hip): real weights in
        row-major or tiled fp32 layout, at most 64 folds (kTail16MaxFolds) of 32 units."""
        ls = [self.layers[j] for j in self._tail]
        lay = next((l._w_layout for l in ls if l.num_output_units == 32), capi.CK_W_ROWMAJOR)
        return (lay in (capi.CK_W_ROWMAJOR, capi.CK_W_TILED_F32) and len(ls) <= 15
                and sum(l.num_folds for l in ls) <= 64 and all(l.arity <= 4 and l.num_input_units == 32 for l in ls))

    def _poison_in_tail(self) -> bool:
        """Whether the tail launch turns the input-validation flag into NaN outputs itself (it writes every circuit
        output: all outputs are few-unit folds of tail layers); otherwise one `ck_poison_outputs` per output follows."""
        if not (self.validate_inputs and self._int_input and self._tail and self._tail16_ok()):
            return False
        return all(int(p) in self._tail and self.layers[int(p)].num_output_units < 32 for p in self._out_pairs[:, 0])

    def _tail_fuses_ll(self) -> bool:
        """Whether `log_likelihood_sum`'s reduction is part of the tail launch (the circuit output is the scalar root)."""
        if not self._tail or not self._tail16_ok() or len(self._out_pairs) != 1 or self._signed:
            return False
        last = self._tail[-1]
        return (int(self._out_pairs[0, 0]) == last and self.layers[last].num_folds == 1
                and self.layers[last].num_output_units == 1)

    def _launch_tail(self, bd: _Binding, stream: int, *, with_ll: bool = False) -> None:
        """One launch for the trailing few-fold layers (cirkit_amd/csrc/ck_tail16.hip, ck_tailp.hip)."""
        n = len(self._tail)
        ls = [self.layers[j] for j in self._tail]
        for l in ls:
            if l._w.is_complex():
                raise ValueError("complex weights in the fused tail")
        lay = next((l._w_layout for l in ls if l.num_output_units == 32), capi.CK_W_ROWMAJOR)
        if self._tail16_ok() and bd.params_at_end:
            keep = self.keep_layer_outputs and (not with_ll or self.keep_levels)  # (a training forward keeps them for the backward)
            desc_dev, levels_dev, n_folds, scratch, ticket, lay = self._tail16_tables(bd, keep=keep, slots=True)
            fuse_ll = with_ll and self._tail_fuses_ll()
            il = self._tailp
            d = capi.TailParamsLaunch()
            d.folds, d.level_begin, d.n_folds, d.n_levels = desc_dev.data_ptr(), levels_dev.data_ptr(), n_folds, n
            d.n_slots, d.B, d.w_layout = self._tail_slots()[2], bd.B, lay
            d.ll = bd.ll.data_ptr() if fuse_ll else None
            d.ll_partial = scratch.data_ptr() if fuse_ll else None
            d.ll_ticket = ticket.data_ptr() if fuse_ll else None
            if fuse_ll and self._recording:  # `log_likelihood_sum(x, out=row)`: the pair goes where input cell 1 points
                d.ll_cell = 1
                bd.ll_cell = 1
            d.bad_input = self._bad_input.data_ptr() if (self._poison_in_tail() and not bd.direct) else None
            t = il["table"]
            d.cat_logits, d.dense_logits = t["src"].data_ptr(), t["dense"].data_ptr()
            d.cat_idx = None if t["idx"] is None else t["idx"].data_ptr()
            d.table, d.table_scale = t["dst"].data_ptr(), t["scale"].data_ptr()
            d.n_tables, d.C = int(t["dense"].shape[0]), int(t["src"].shape[2])
            d.rows, d.n_rows = il["rows_all"].data_ptr(), il["n_rows_all"]
            capi.call("ck_tail_params_fwd", C.byref(d), stream)
            return
        if self._tail16_ok():
            # `log_likelihood_sum` returns [sum, count] only: the tail's inner folds stay in LDS; `forward` keeps the layer
            # outputs (`layer_outputs()` reads them) unless the caller opted out
            keep = self.keep_layer_outputs and (not with_ll or self.keep_levels)  # (a training forward keeps them for the backward)
            desc_dev, levels_dev, n_folds, scratch, ticket, lay = self._tail16_tables(bd, keep=keep)
            fuse_ll = with_ll and self._tail_fuses_ll()
            capi.call(
                "ck_tail16_lse_fwd", desc_dev.data_ptr(), n_folds, levels_dev.data_ptr(), n, bd.B, 32, lay,
                bd.ll.data_ptr() if fuse_ll else None, scratch.data_ptr() if fuse_ll else None,
                ticket.data_ptr() if fuse_ll else None,
                self._bad_input.data_ptr() if (self._poison_in_tail() and not bd.direct) else None,
                1 if self._signed else 0, stream,
            )
            return
        raise capi.HipExtensionError("a fused tail that does not fit the 16-row walk (cirkit_amd/fusion.py find_tail only proposes tails that do)")

    def _tail16_tables(self, bd: _Binding, *, keep: bool = True, slots: bool = False) -> tuple:
        """(fold descriptors, level table, number of folds, per-tile LL sums, LL ticket, weight layout) of the 16-row tail
        walk -- `ck_tail16_lse_fwd`, or the tail phase of the leaf launch -- for this binding.  keep=False: 32-unit folds
        that only the tail itself reads are not stored (they are not circuit outputs and no later launch reads them)."""
        ls = [self.layers[j] for j in self._tail]
        lay = next((l._w_layout for l in ls if l.num_output_units == 32), capi.CK_W_ROWMAJOR)
        key = ("tail16" if keep else "tail16-nokeep") + ("-slots" if slots else "")
        slot_of = self._tail_slots()[0] if slots else None
        tabs = bd.cp_tabs.get(key)
        if tabs is None:
            first, acc = {}, 0
            for j, l in zip(self._tail, ls):
                first[j] = acc
                acc += l.num_folds
            desc = np.zeros(acc, dtype=_TAIL16_FOLD)
            outs = {int(p) for p in self._out_pairs[:, 0]}
            read_later = {int(p) for jj, ch in enumerate(self._children) if ch is not None and jj not in self._tail
                          for p in np.unique(ch[..., 0])}
            arena = bd.arena.data_ptr()
            esz = 8 if self._signed else 4  # (signed: complex64 blocks)
            for j, l in zip(self._tail, ls):
                ch = self._children[j]  # (F, H, 2): producer layer, fold
                off = bd.row_off[j].cpu().numpy()
                Ko = l.num_output_units
                for f in range(l.num_folds):
                    d = desc[first[j] + f]
                    d["w"] = l._w.data_ptr() + f * Ko * 32 * 4
                    d["out"] = bd.views[j].data_ptr() + f * bd.B * Ko * esz
                    d["H"], d["Ko"] = l.arity, Ko
                    d["skip_store"] = 0 if (keep or Ko != 32 or j in outs or j in read_later) else 1
                    d["slot"] = (slot_of.get((j, f), 0) if slots else first[j] + f)
                    d["child_src"][:] = -1
                    for h in range(l.arity):
                        pj, pf = int(ch[f, h, 0]), int(ch[f, h, 1])
                        if pj in first and self.layers[pj].num_output_units == 32:
                            d["child_src"][h] = slot_of[(pj, pf)] if slots else first[pj] + pf
                        d["child"][h] = arena + int(off[f, h]) * esz
            levels = np.asarray([first[j] for j in self._tail] + [acc], dtype=np.int32)
            shared = next((bd.cp_tabs[k] for k in ("tail16", "tail16-nokeep", "tail16-slots", "tail16-nokeep-slots")
                           if k in bd.cp_tabs), None)  # (one LL scratch / ticket per binding)
            tabs = bd.cp_tabs[key] = (
                torch.from_numpy(desc.view(np.uint8)).to(self.device), torch.from_numpy(levels).to(self.device), acc,
                shared[3] if shared else torch.zeros((bd.B + 15) // 16 + 1, dtype=torch.float64, device=self.device),
                shared[4] if shared else torch.zeros(1, dtype=torch.int32, device=self.device))
        return (*tabs, lay)

    def _group_table(self, g: SubtreeGroup, stream: int | None):
        """The (table, in-kernel dense weight) pair a fused leaf launch reads.  With `dense_on_table`
        the dense layer is pushed through the table first: T'[d] = dense_d(table[leaf(d)]) over the
        C categories (+ the integral row); `stream` None only looks the buffers up."""
        dev = self._group_dev.get(g.root)
        if dev is None:
            dev = (torch.from_numpy(g.nodes).to(self.device),)
            self._group_dev[g.root] = dev
        cat = self.layers[g.input_layer]
        w_dense = None if g.dense_layer is None else self.layers[g.dense_layer]._w
        if g.root in self._table_fused:  # built by the prologue (kind-4 job)
            return dev[1], None
        if w_dense is None or not self.dense_on_table or g.depth == 0:
            return cat._table, w_dense
        dl = self.layers[g.dense_layer]
        Cn, K = cat.num_categories, cat.num_output_units
        if len(dev) == 1:
            leaf_of_dense = self._children[g.dense_layer][:, 0, 1].astype(np.int64)
            dev = dev + (
                torch.empty((dl.num_folds, Cn + 1, K), dtype=torch.float32, device=self.device),
                torch.from_numpy(np.ascontiguousarray(leaf_of_dense * ((Cn + 1) * K))).to(self.device),
            )
            self._group_dev[g.root] = dev
        if stream is not None:
            capi.call(
                "ck_sum_lse_fwd", cat._table.data_ptr(), dev[2].data_ptr(), w_dense.data_ptr(), dev[1].data_ptr(),
                dl.num_folds, 1, Cn + 1, K, K, capi.CK_SUM_CAT, dl._w_layout, stream,
            )
        return dev[1], None

    def _leaf_is_persistent(self, g: SubtreeGroup, B: int) -> bool:
        """Whether the fused leaf launch of group g at batch size B is the persistent one (ck_leaf.hip)."""
        cat = self.layers[g.input_layer]
        if (self.persistent_leaf is False or g.root not in self._table_fused or not self.linear_levels or g.depth < 1
                or cat.num_output_units != 32 or cat.num_categories >= 65535
                or self._group_layout(g) not in (capi.CK_W_TILED_F32, capi.CK_W_ROWMAJOR) or self.plan.num_variables * B >= 2**31):
            return False
        return self.persistent_leaf is True or self.layers[g.root].num_folds * ((B + 31) // 32) >= self._n_cu

    def _launch_group(self, g: SubtreeGroup, bd: _Binding, out: torch.Tensor, stream: int, *, with_table: bool = False,
                      with_ll: bool = False) -> None:
        """One fused launch for Categorical -> [dense] -> CP-T levels (cirkit_amd/csrc/ck_fused.hip)."""
        if self._signed:
            return self._launch_group_signed(g, bd, out, stream)
        table, w_dense = self._group_table(g, stream if with_table else None)
        dev = self._group_dev[g.root]
        cat = self.layers[g.input_layer]
        levels = (C.c_void_p * max(1, g.depth))(*[self.layers[j]._w.data_ptr() for j in g.levels])
        node_off = (C.c_int32 * (g.depth + 1))(*g.node_off)
        scale = dev[3] if g.root in self._table_fused and len(dev) > 3 else None
        F_root, n_tiles = self.layers[g.root].num_folds, (bd.B + 31) // 32
        persistent = scale is not None and w_dense is None and self._leaf_is_persistent(g, bd.B)
        if persistent:
            work = bd.cp_tabs.get((g.root, "leaf_work"))
            if work is None:
                work = bd.cp_tabs[(g.root, "leaf_work")] = torch.from_numpy(
                    leaf_segments(F_root, n_tiles, self._n_cu)).to(self.device)
            self._leaf_walk_root = g.root
            self._leaf_walk_pairs = self._leaves_in_adjacent_pairs(g)
            self._leaf_walk(bd, table=table, scale=scale, scope=cat._scope(self.device), levels=levels, nodes=dev[0],
                            node_off=node_off, leaf_off=g.leaf_off, out=out, work=work, depth=g.depth,
                            K=cat.num_output_units, Cn=cat.num_categories, w_layout=self._group_layout(g), redo=None,
                            n_roots=F_root, waves=8, stream=stream, keep=self._keep_buffers(g, bd))
            return
        capi.call(
            "ck_subtree_cat_cpt_fwd", table.data_ptr(), None if scale is None else scale.data_ptr(), bd.xt_i.data_ptr(),
            cat._scope(self.device).data_ptr(),
            None if w_dense is None else w_dense.data_ptr(), levels, dev[0].data_ptr(), node_off, g.leaf_off,
            out.data_ptr(), g.depth, self.layers[g.root].num_folds, bd.B, cat.num_output_units,
            cat.num_categories, self._group_layout(g), stream,
        )

    def _keep_buffers(self, g: SubtreeGroup, bd: _Binding):
        """`keep_levels`: ([(F_l, tiles, 1024) tile-native per fused level l = 2, 4 -- None for the levels in between, which the
        backward recomputes], (F_root, tiles) int32 flags) of group g in this binding, else None."""
        if not self.keep_levels:
            return None
        hit = bd.keep.get(g.root)
        if hit is None:
            tiles = (bd.B + 31) // 32
            hit = bd.keep[g.root] = (
                [torch.empty((self.layers[j].num_folds, tiles, 1024), dtype=torch.float32, device=self.device) if l % 2 == 1 else None
                 for l, j in enumerate(g.levels)],
                torch.zeros(self.layers[g.root].num_folds * tiles, dtype=torch.int32, device=self.device))
        return hit

    def _launch_group_signed(self, g: SubtreeGroup, bd: _Binding, out: torch.Tensor, stream: int) -> None:
        """Embedding -> CP-T levels of a real-valued complex circuit: the persistent leaf launch on signed linear tiles
        (the Embedding weight table IS the linear table, scale 0) followed by its marked-tile launch."""
        emb = self.layers[g.input_layer]
        dev = self._group_dev.get(g.root)
        if dev is None or len(dev) < 2:
            nodes = dev[0] if dev else torch.from_numpy(g.nodes).to(self.device)
            dev = self._group_dev[g.root] = (
                nodes, torch.zeros((emb.num_folds, emb.num_states + 1), dtype=torch.float32, device=self.device))
        levels = (C.c_void_p * g.depth)(*[self.layers[j]._w.data_ptr() for j in g.levels])
        node_off = (C.c_int32 * (g.depth + 1))(*g.node_off)
        F_root, n_tiles = self.layers[g.root].num_folds, (bd.B + 31) // 32
        work = bd.cp_tabs.get((g.root, "leaf_work"))
        if work is None:
            work = bd.cp_tabs[(g.root, "leaf_work")] = (
                torch.from_numpy(leaf_segments(F_root, n_tiles, self._n_cu)).to(self.device),
                torch.zeros(F_root * n_tiles, dtype=torch.int32, device=self.device))
        segs, redo = work
        self._leaf_walk_root = g.root
        self._leaf_walk_pairs = self._leaves_in_adjacent_pairs(g)
        self._leaf_walk(bd, table=emb._table, scale=dev[1], scope=emb._scope(self.device), levels=levels, nodes=dev[0],
                        node_off=node_off, leaf_off=g.leaf_off, out=out, work=segs, depth=g.depth, K=emb.num_output_units,
                        Cn=emb.num_states, w_layout=self._group_layout(g), redo=redo, n_roots=F_root, waves=8, stream=stream)

    def _leaves_in_adjacent_pairs(self, g: SubtreeGroup) -> bool:
        """Leaves 2j and 2j + 1 of every root of the group read variables v and v + 1, v even (what region graphs over
        images give): the raw batch is then fetched with one 16-byte load per pair of leaves."""
        hit = self._group_dev.get(("pairs", g.root))
        if hit is None:
            kl = 1 << g.depth
            leaf_ids = np.asarray(g.nodes[g.leaf_off:g.leaf_off + self.layers[g.root].num_folds * kl]).reshape(-1, kl)
            var = self.layers[g.input_layer].scope_idx[:, 0][leaf_ids]
            hit = bool(kl >= 4 and self.plan.num_variables % 2 == 0 and np.all(var[:, 0::2] % 2 == 0)
                       and np.all(var[:, 1::2] == var[:, 0::2] + 1))
            self._group_dev[("pairs", g.root)] = hit
        return hit

    def _leaf_root_table(self, nodes: torch.Tensor, node_off, leaf_off: int, scope: torch.Tensor, depth: int, n_roots: int) -> torch.Tensor:
        """(roots, 3 * 2^depth) int32 on the device (`ck_leaf_launch.root_tab`): per root of a fused leaf region the variable
        and the table fold of each of its leaves, then the folds of its nodes in the order of the walk's steps (leaf i is
        followed by as many steps as i has trailing one bits: level l + 1 takes fold nodes[node_off[l + 1] + ...])."""
        key = ("root_tab", nodes.data_ptr(), int(leaf_off), depth)
        hit = self._group_dev.get(key)
        if hit is None:
            nd = nodes.cpu().numpy().astype(np.int64)
            sc = scope.cpu().numpy().astype(np.int64)
            off = [int(v) for v in node_off[: depth + 1]]
            kl = 1 << depth
            tab = np.zeros((n_roots, 3 * kl), dtype=np.int32)
            for t in range(n_roots):
                tab[t, :kl] = sc[nd[leaf_off + t * kl: leaf_off + (t + 1) * kl]]
                tab[t, kl:2 * kl] = nd[off[0] + t * kl: off[0] + (t + 1) * kl]
                k = 0
                for i in range(kl):
                    l = 0
                    while (i >> l) & 1:  # the steps behind leaf i: levels 1, 2, ... while the bits of i are set
                        tab[t, 2 * kl + k] = nd[off[l + 1] + t * (kl >> (l + 1)) + (i >> (l + 1))]
                        k += 1
                        l += 1
            hit = self._group_dev[key] = torch.from_numpy(tab).to(self.device)
        return hit

    def _leaf_walk(self, bd: _Binding, *, table, scale, scope, levels, nodes, node_off, leaf_off, out, work, depth, K, Cn,
                   w_layout, redo, n_roots, waves, stream, keep=None) -> None:
        """`ck_leaf_walk_fwd`: the persistent leaf launch over the staged batch or -- `bd.direct` -- over the caller's."""
        d = capi.LeafLaunch()
        d.table, d.table_scale, d.scope = table.data_ptr(), scale.data_ptr(), scope.data_ptr()
        d.w_levels, d.nodes, d.node_off, d.leaf_off = levels, nodes.data_ptr(), node_off, leaf_off
        d.out, d.work, d.n_seg, d.n_wg, d.waves, d.depth = out.data_ptr(), work.data_ptr(), int(work.shape[0]), self._n_cu, waves, depth
        d.B, d.K, d.C, d.w_layout = bd.B, K, Cn, w_layout
        d.contraction = {"f32": 0, "bf16x3": 3, "bf16x6": 6}[self.contraction]
        if d.contraction and not (bd.direct and depth == 4 and redo is None and keep is None):
            raise ValueError(f"contraction={self.contraction!r} is a variant of the depth-4 persistent leaf launch over the caller's batch "
                             "(unsigned values, inference forward); this circuit / batch does not take that launch")
        d.signed_redo, d.n_roots = (None if redo is None else redo.data_ptr()), n_roots
        d.root_tab = self._leaf_root_table(nodes, node_off, leaf_off, scope, depth, n_roots).data_ptr()
        if bd.direct:
            d.xt, d.preclamped, d.D = None, 0, self.plan.num_variables
            d.x_rows, d.x_input = self._raw_batch_args(bd)
            d.bad_input = self._bad_input.data_ptr() if self.validate_inputs else None
            d.x_pairs = 1 if getattr(self, "_leaf_walk_pairs", False) else 0
        else:
            d.xt, d.preclamped, d.x_rows, d.x_input = bd.xt_i.data_ptr(), (1 if self._preclamp() else 0), None, -1
        if keep is not None:
            if not bd.direct:
                raise ValueError("keep_levels needs the leaf launch to read the caller's batch (direct_input)")
            d.keep_levels = (C.c_void_p * depth)(*[None if t is None else t.data_ptr() for t in keep[0]])
            d.keep_redo = keep[1].data_ptr()
        capi.call("ck_leaf_walk_fwd", C.byref(d), stream)
