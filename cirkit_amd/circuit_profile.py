"""Instrumentation of a `HipCircuit` (split out of circuit.py; `HipCircuit` inherits this mixin): which HIP kernel
evaluates a layer (`kernel_label`, the names of a rocprofv3 trace) and an instrumented eager pass with HIP events around
every launch (`profile_kernels`: per launch its time, the SURVEY.md 8(d) algorithmic bytes / flops of the reference layers
it stands for and the contraction flops it executes) -- what bench.py's `roofline` and scripts/bench_plan.py read.
"""

from __future__ import annotations


import numpy as np
import torch

from . import _capi as capi
from .layers import HipConstantValueLayer, HipInputLayer


class _ProfilingMixin:
    # -- instrumentation -------------------------------------------------------------------------
    def kernel_label(self, i: int, B: int = 4096) -> str:
        """Name of the HIP kernel that evaluates layer i (as it appears in a rocprofv3 trace)."""
        l, s = self.layers[i], self.plan.layers[i]
        if i in self._input_prod:
            return "gaussian_prod_rows16_kernel" if (B % 4 == 0 and l.num_output_units in (32, 64, 128, 256)) else "gaussian_prod_kernel<8>"
        if i in self._tdense:
            return "gather_rows_vec (dense layer tabulated over its categories)"
        if i in self._emb_gather:
            return "sum_clse_tile32 (Embedding rows gathered from the table)"
        def gathers(slot_dense) -> bool:  # some slot reads a tabulated dense layer
            return any(int(d) in self._tdense for d in np.unique(slot_dense[..., 0]) if d >= 0)

        # (ck_cp.hip: region_dma_kernel<NK, WAVES, MINW, LINEAR, BLOCK, CT>; bf16x6 at K = 64: two workgroups per CU)
        dma = ("2, 4, 2" if self._ct == 6 else "2, 4, 3") if l.num_output_units == 64 else "1, 8, 2"
        if i in self._regions:
            if gathers(self._regions[i].slot_dense):
                return "region_lse_kernel<2, 4, 3>" if l.num_output_units == 64 else "region_lse_kernel<1, 8, 4>"
            return f"region_dma_kernel<{dma}, {'true' if self.linear_levels else 'false'}, false, {self._ct}>"
        if i in self._cp_blocks and self._cp_subset.get(i) is None and (
                self._cp_blocks[i].slot_dense.shape[1] <= 8 or not gathers(self._cp_blocks[i].slot_dense)):
            return f"region_dma_kernel<{dma}, false, true, {self._ct}>"
        if i in self._cp_blocks or i in self._cp_leftover:
            return f"cp_lse_kernel<{l.num_output_units // 32}, 8, {'true' if i in self._cp_blocks else 'false'}>"
        if i in self._group_of_root and self._signed:
            raw = "true" if self._direct_input(B) else "false"
            g = self._group_of_root[i]
            xp = "true" if (raw == "true" and g.depth >= 2 and self._leaves_in_adjacent_pairs(g)) else "false"
            return f"leaf_persistent_kernel<{g.depth}, 8, true, {raw}, {xp}, false, 0> (signed: real-valued complex circuit)"
        if i in self._group_of_root:
            g = self._group_of_root[i]
            in_kernel_dense = g.dense_layer is not None and not (self.dense_on_table and g.depth > 0)
            if i in self._table_fused and self.linear_levels:
                if self._leaf_is_persistent(g, B):
                    raw = "true" if self._direct_input(B) else "false"
                    xp = "true" if (raw == "true" and g.depth >= 2 and self._leaves_in_adjacent_pairs(g)) else "false"
                    ct = {"f32": 0, "bf16x3": 3, "bf16x6": 6}[self.contraction]
                    return f"leaf_persistent_kernel<{g.depth}, 8, false, {raw}, {xp}, {'true' if self.keep_levels else 'false'}, {ct}>"
                return f"subtree_linear_kernel<{g.depth}, {self._group_layout(g)}>"
            return (f"subtree_cat_cpt_kernel<{g.depth}, {'true' if in_kernel_dense else 'false'}, "
                    f"{self._group_layout(g)}>")
        if s.type in ("categorical", "embedding", "binomial"):
            return "gather_rows_vec" if l.num_output_units % 4 == 0 else "gather_rows_scalar"
        if s.type == "gaussian":
            return "gaussian_kernel"
        if s.type == "constant":
            return "constant_kernel"
        if s.type == "hadamard":
            return "hadamard_vec" if (l.num_input_units * l.esize) % 4 == 0 else "hadamard_scalar"
        if s.type == "kronecker":
            return "kronecker_kernel"
        if s.type == "tensordot":
            return "tensordot_lse_kernel"
        if getattr(l, "_mixing", False):
            k4 = l.num_output_units // 4
            vec = l.num_output_units % 4 == 0 and 1 <= k4 <= 64 and (k4 & (k4 - 1)) == 0
            return "mixing_lse_vec" if vec else "mixing_lse_kernel"
        prod_like = s.type == "cpt" or l.arity == 1
        if (self._complex and prod_like and s.type in ("sum", "cpt") and l.num_input_units == l.num_output_units == 32
                and l._w is not None and not l._w.is_complex()):
            return "sum_clse_tile32"
        if (not self._complex and prod_like and l.num_input_units == l.num_output_units
                and l.num_input_units in (32, 64)):
            return f"sum_lse_tile32<{l._w_layout}>" if l.num_input_units == 32 else "cp_lse_kernel<2, 8, false>"
        if (not self._complex and s.type == "sum" and l.arity > 1 and l.num_input_units == l.num_output_units
                and l.num_input_units in (32, 64)):
            nk = l.num_input_units // 32
            waves = 4 if nk == 2 else 8
            if (2 * 32 + waves * 32 + l.arity) * l.num_input_units * 4 <= 80 * 1024:  # (ck_cp.hip cat_dense)
                return f"region_dma_kernel<{nk}, {waves}, {3 if nk == 2 else 2}, false>"
            return f"cat_lse_kernel<{nk}, 8>"
        if not self._complex and s.type in ("sum", "cpt") and not getattr(l, "_mixing", False):
            cat = s.type == "sum" and l.arity > 1
            n = l.num_input_units * (l.arity if cat else 1)
            if l.num_input_units % 32 == 0 and l.num_output_units % 32 == 0 and 32 <= n <= 256:
                return f"sum_lse_gemm_kernel<{n // 32}, {'true' if cat else 'false'}>"
            if l.num_input_units % 32 == 0 and l.num_output_units % 32 == 0 and (
                    (256 < n <= 512 and n % 64 == 0) or n in (768, 1024)):
                sp = 2 if n <= 512 else 4
                return f"sum_lse_gemm_split_kernel<{n // 32 // sp}, {sp}, {'true' if cat else 'false'}>"
        if not self._complex and s.type == "tucker" and l.arity == 2 and l.num_input_units in (32, 64):
            wg1 = l.num_folds * ((l.num_output_units + 31) // 32) * ((B + 127) // 128)
            # (ck_gemm.hip tucker_lse: few tiles per resident slot; the bf16 variants take the stream-K launch at any size)
            if (wg1 <= 8 * 3 * self._n_cu or self._ct) and self._scratch() is not None:
                logits = "true" if getattr(l, "_use_logits", False) or (l._logits_ok and l._theta is not None) else "false"
                return f"tucker_streamk_kernel<{l.num_input_units // 32}, {logits}, {self._ct}, {4 if self._ct else 1}, {2 if self._ct else 3}>"
            return f"tucker_lse_kernel<{l.num_input_units // 32}>"
        if (not self._complex and s.type in ("sum", "cpt") and not getattr(l, "_mixing", False) and (s.type == "cpt" or l.arity == 1)
                and l.num_output_units <= 4 and l.num_input_units in (32, 64)):
            return f"sum_lse_few_outputs_kernel<{l.num_input_units}>"  # (the scalar folds at the top of a circuit)
        return "sum_lse_generic"

    def profile_kernels(self, x: torch.Tensor | None, iters: int = 10) -> list[dict]:
        """Eager (non-graph) forwards with HIP events around every layer's parameter kernels and
        layer kernel, recorded on the current stream (the stream the kernels are launched on).
        Returns one row per launch group: kernel label, mean ms, algorithmic bytes / flops (those of the
        reference layers the launch stands for) and `executed_flops` (the contraction flops the launch itself issues: a
        dense layer pushed through its category table is executed by the prologue on C + 1 rows, not by the leaf launch
        on B rows)."""
        with torch.cuda.device(self.device):
            return self._profile_kernels(x, iters)

    def _profile_kernels(self, x: torch.Tensor | None, iters: int) -> list[dict]:
        bd = self._run(x)  # make sure the binding (arena, staging copy) exists and is warm
        B = bd.B
        if getattr(self, "_clin", None) is not None:  # (the batch staged by `_run` above is still there)
            return self._clin.profile(bd, iters)
        cur = torch.cuda.current_stream(self.device)
        stream = cur.cuda_stream
        esz = 8 if self._complex else 4
        rows: list[dict] = []
        acc: list[list[float]] = []
        ws = self._scratch_for(B)
        if ws is not None:  # (as `_enqueue_layers` does: the launches below are the ones a forward records)
            capi.call("ck_set_workspace", ws.data_ptr(), ws.numel() * 4)
        stage_ms: list[float] = []
        xf_xi = self._prepare_input(x) if self.plan.num_variables else (None, None)
        for it in range(iters + 1):
            evs = []
            try:  # keep the GPU busy while the host enqueues, so the events bracket GPU time only
                torch.cuda._sleep(4_000_000)
            except Exception:  # pragma: no cover
                pass
            if self.plan.num_variables and not (bd.direct and xf_xi[0] is None):  # the staging launch(es) of a forward
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record(cur)
                self._stage_input(bd, xf_xi[0], None if bd.direct else xf_xi[1], stream)
                s1.record(cur)
                stage_ms.append((s0, s1))
            for i, (l, view, ro) in enumerate(zip(self.layers, bd.views, bd.row_off)):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e2 = torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                if i == 0:
                    self._enqueue_params_batch_only(stream, bd)
                in_tail = bool(self._tail) and i in self._tail
                if in_tail and i == self._tail[0]:
                    for j in self._tail:
                        self.layers[j].prepare(stream, batched=self.batch_params)
                elif not in_tail:
                    l.prepare(stream, batched=self.batch_params)
                if i in self._group_of_root:  # the dense layer pushed through the table is parameter-side work
                    self._group_table(self._group_of_root[i], stream)
                e1.record(cur)
                if in_tail:
                    if i == self._tail[0]:
                        self._launch_tail(bd, stream)
                elif i in self._virtual or i in self._td_first:
                    pass
                elif i in self._td_had or i in self._td_pair:
                    self._launch_tensordot(i, bd, stream)
                elif i in self._group_of_root:
                    self._launch_group(self._group_of_root[i], bd, view, stream)
                elif i in self._tdense:
                    self._launch_table_dense(i, bd, stream)
                elif i in self._emb_gather:
                    self._launch_emb_gather(i, bd, stream)
                elif i in self._cp_blocks or i in self._cp_leftover:
                    self._launch_cp(i, bd, stream)
                elif i in self._regions:
                    self._launch_region(i, bd, stream)
                elif i in self._input_prod:
                    self._launch_input_prod(i, bd, stream)
                elif isinstance(l, HipConstantValueLayer):
                    l.launch_const(view, B, stream)
                elif isinstance(l, HipInputLayer):
                    l.launch_input(bd.xt if l.wants_float_input else bd.xt_i, self.plan.num_variables, view, B, stream)
                else:
                    l.launch(bd.arena, ro, view, B, stream)
                e2.record(cur)
                evs.append((e0, e1, e2))
            torch.cuda.synchronize(self.device)
            if it == 0:
                continue  # warm-up
            acc.append([t for e0, e1, e2 in evs for t in (e0.elapsed_time(e1), e1.elapsed_time(e2))])
        if ws is not None:
            capi.call("ck_set_workspace", None, 0)
        # an event pair with nothing between still measures a few us of marker overhead: it is
        # calibrated on empty pairs and subtracted; intervals without a launch are dropped below
        # (`has_prep` / virtual layers)
        try:
            torch.cuda._sleep(4_000_000)
        except Exception:  # pragma: no cover
            pass
        empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(32)]
        for a, b in empty:
            a.record(cur)
            b.record(cur)
        torch.cuda.synchronize(self.device)
        overhead = float(np.median([a.elapsed_time(b) for a, b in empty]))
        mean = np.maximum(np.mean(np.asarray(acc), axis=0) - overhead, 0.0)
        if stage_ms:
            ms = max(float(np.mean([a.elapsed_time(b) for a, b in stage_ms[1:]])) - overhead, 0.0)
            rows.append({"layer": -1, "kernel": "stage_categories_kernel" if self._int_input else "transpose_kernel<float, float>",
                         "ms": ms, "algorithmic_bytes": float(self.plan.num_variables * B * 8)})
        layer_bytes: dict[int, float] = {}
        layer_flops: dict[int, float] = {}
        moved = [0.0]  # flops of dense layers evaluated on their category tables by the prologue
        for i, (l, s) in enumerate(zip(self.layers, self.plan.layers)):
            pbytes = 0
            for pg in s.params.values():
                for n in pg.nodes:
                    if n.op in ("tensor", "pointer"):
                        shp, dt = self.plan.tensors[n.config["tensor"]]
                        per_fold = int(np.prod(shp[1:])) * (8 if "complex" in dt else 4)
                        pbytes += per_fold * n.num_folds
            has_prep = bool(s.params) and not (self.batch_params and l._batched)
            if i == 0 and self.batch_params and self._batch is not None and len(self._batch) and not (
                    bd.params_at_end and self._tailp["rest"] is None):
                rows.append({"layer": 0, "kernel": "softmax_batch_kernel<false>", "ms": float(mean[0]),
                             "algorithmic_bytes": float(2 * sum(
                                 int(np.prod(shp)) * 4 for shp, _ in self.plan.tensors.values()))})
            elif has_prep:
                rows.append({"layer": i, "kernel": "param kernels (per node)", "ms": float(mean[2 * i]),
                             "algorithmic_bytes": float(pbytes)})
            if i in self._group_of_root:
                g = self._group_of_root[i]
                if g.dense_layer is not None and self.dense_on_table and g.depth > 0 and i not in self._table_fused:
                    cat, dl = self.layers[g.input_layer], self.layers[g.dense_layer]
                    tb = 2.0 * dl.num_folds * (cat.num_categories + 1) * cat.num_output_units * 4
                    rows.append({"layer": i, "kernel": f"sum_lse_tile32<{dl._w_layout}> (dense layer on the table)",
                                 "ms": float(mean[2 * i]), "algorithmic_bytes": tb})
            if s.inputs is not None:
                rd = l.num_folds * l.arity * B * l.num_input_units * esz
            elif s.scope_idx is not None and s.scope_idx.size:
                rd = int(s.scope_idx.size) * B * 8
            else:
                rd = 0
            wr = l.num_folds * B * l.num_output_units * esz
            layer_bytes[i] = float(rd + wr)
            if s.type in ("sum", "cpt", "tensordot") and not getattr(l, "_mixing", False):
                n_in = l.num_input_units * (l.arity if s.type == "sum" else 1)
                if s.type == "tensordot":
                    n_in = l._num_contract_units
                layer_flops[i] = 2.0 * l.num_folds * B * l.num_output_units * n_in * (4 if self._complex else 1)
            else:
                layer_flops[i] = 0.0
            if i in self._virtual:
                continue
            if self._tail and i in self._tail:
                if i == self._tail[-1]:
                    tl = next((self.layers[j]._w_layout for j in self._tail
                               if self.layers[j].num_output_units == 32), 0)
                    rows.append({"layer": self._tail[0], "kernel": ("tail_params_kernel" if bd.params_at_end else f"tail16_kernel<{tl}, {'true' if self._signed else 'false'}>"),
                                 "ms": float(mean[2 * self._tail[0] + 1]),
                                 "algorithmic_bytes": sum(layer_bytes[j] for j in self._tail),
                                 "algorithmic_flops": sum(layer_flops[j] for j in self._tail)})
                continue
            nbytes, nflops = layer_bytes[i], layer_flops[i]
            if i in self._input_prod:
                nbytes += layer_bytes[self._input_prod[i]]
            if i in self._emb_gather:
                nbytes += layer_bytes[self._emb_gather[i]]
            if i in self._cp_leftover:  # only the folds other consumers need are evaluated here
                share = len(self._cp_leftover[i]) / l.num_folds
                nbytes, nflops = nbytes * share, nflops * share
            if i in self._cp_blocks:  # plus the dense folds evaluated inside the launch
                sub = self._cp_subset.get(i)
                nb, nf = self._cp_fold_cost(i, np.arange(l.num_folds) if sub is None else sub, layer_bytes, layer_flops)
                nbytes, nflops = nb, nf
            if i in self._regions:  # plus the CP blocks (and their dense folds) it takes over
                ch = self._children[i]
                for h in np.unique(ch[..., 0]):
                    nb, nf = self._cp_fold_cost(int(h), ch[..., 1][ch[..., 0] == h], layer_bytes, layer_flops)
                    nbytes += nb
                    nflops += nf
            executed = nflops
            if i in self._group_of_root:  # the fused launch does the work of every layer it replaces
                g = self._group_of_root[i]
                nbytes += sum(layer_bytes[j] for j in g.virtual)
                nflops += sum(layer_flops[j] for j in g.virtual)
                executed = nflops
                if g.dense_layer is not None and (i in self._table_fused or (self.dense_on_table and g.depth > 0)):
                    # the dense layer is evaluated on the (C + 1)-row table by the parameter prologue, not by this launch:
                    # its flops are part of what the launch stands for (algorithmic) but not of what it executes
                    executed -= layer_flops[g.dense_layer]
                    Cn = self.layers[g.input_layer].num_categories
                    moved[0] += layer_flops[g.dense_layer] * (Cn + 1) / B
            rows.append({"layer": i, "kernel": self.kernel_label(i, B), "ms": float(mean[2 * i + 1]),
                         "algorithmic_bytes": nbytes, "algorithmic_flops": nflops, "executed_flops": executed})
        for r in rows:  # the prologue executes the dense layers that were pushed through their tables
            if r["kernel"].startswith("softmax_batch_kernel"):
                r["executed_flops"] = r.get("executed_flops", 0.0) + moved[0]
            r.setdefault("executed_flops", r.get("algorithmic_flops", 0.0))
        return rows

    def _cp_fold_cost(self, i: int, folds: np.ndarray, layer_bytes, layer_flops) -> tuple[float, float]:
        """Algorithmic bytes / flops (reference layer boundaries) of `folds` of CP-block layer i,
        including the dense folds evaluated inside them."""
        nb = layer_bytes[i] * len(folds) / self.layers[i].num_folds
        nf = layer_flops[i] * len(folds) / self.layers[i].num_folds
        dl = self._cp_blocks[i].slot_dense[folds][..., 0]
        for d in np.unique(dl):
            if d >= 0:
                share = float((dl == d).sum()) / self.layers[int(d)].num_folds
                nb += share * layer_bytes[int(d)]
                nf += share * layer_flops[int(d)]
        return nb, nf
