"""Training step on the HIP backend: forward, backward, optimiser -- SURVEY.md section 8 (f3).

The reference trains with autograd through its torch forward (notebooks/learning-a-circuit.ipynb,
cell 18: ``loss = -torch.mean(circuit(batch)); loss.backward(); optimizer.step()``).  Two forms here:

* FUSED (circuits whose leaf region is one persistent launch -- Categorical -> dense -> 2 or 4 CP-T levels, BASELINE
  configs 2 / 3 -- `HipTrainer(fused=None)` picks it when it applies): the forward is the fused inference forward that also
  keeps the linear tile of every node of the leaf region (`ck_leaf_walk_fwd` keep_levels) and the outputs of the tail; the
  backward walks the tail layer by layer, then the leaf region two levels per launch on those tiles (`ck_leaf_walk_bwd`:
  no layer gradient is materialised above the leaves), scatters the leaf gradients into the (F, C + 1, 32) table by
  category, and pushes the table gradient through the dense layer and the Categorical log-softmax ON THE TABLE (C + 1
  rows per fold, like `dense_on_table` in the forward);
* LAYER-WISE (everything else, and the checker of the fused form): the forward with every layer output kept in the arena,
  then a reverse launch list of hand-written kernels (cirkit_amd/csrc/ck_backward.hip) over the plan.

Data-parallel training = one process per GPU, the batch sharded, and ONE all-reduce of a
single flat gradient buffer (all parameter gradients are views of it) over RCCL/xGMI per step.

Covered: real lse-sum circuits made of Categorical (probs = softmax) or Gaussian inputs, Sum / CP-T /
Tucker layers (softmax, raw, or any parameter graph `HipParameter.backward` handles, e.g. the MatMul weight
of a collapsed Sum -> Sum pair), mixing layers, Hadamard and Kronecker layers -- what the image / tabular
templates build with 'cp', 'cp-t' and 'tucker' (BASELINE configs 1-4, the reference's Tucker notebook).  Other layers
(TensorDot, the complex semiring) raise NotImplementedError.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Mapping

import numpy as np
import torch

from .distributed import all_reduce_sum as _all_reduce_sum, default_comm as _default_comm, world_size as _world_size
from . import _capi as capi
from .circuit import HipCircuit
from .layers import (HipCategoricalLayer, HipCPTLayer, HipGaussianLayer, HipHadamardLayer, HipKroneckerLayer, HipSumLayer,
                     HipTuckerLayer)
from .parameters import TensorStore
from .plan import Plan


class HipTrainer:
    """Maximum-likelihood training of a plan's parameters: ``loss = -mean_b log p(x_b)``."""

    def __init__(
        self,
        plan: Plan,
        tensors: Mapping[str, object],
        *,
        device: str | torch.device = "cuda:0",
        lr: float = 0.01,
        optimizer: str = "adam",
        betas: tuple[float, float] = (0.9, 0.999),
        eps: float = 1e-8,
        pad_units: bool = False,
        fused: bool | None = None,
        jobs: bool | None = None,
        fuse_optimizer: bool = True,
    ) -> None:
        """`pad_units`: train the plan with its unit counts padded to multiples of 32 (cirkit_amd/padding.py), so that
        the MFMA forward / backward tiles apply to any width.  The padded entries never receive a gradient (softmax
        at a -inf logit, zero weight on every padded unit), so the padded circuit stays the same function; `self.grads`
        and the parameter store then hold the PADDED tensors -- `gradients()` / `parameters()` return user shapes.
        `fused`: None takes the fused forward / backward when the plan qualifies (module docstring), True insists
        (NotImplementedError says why not), False forces the layer-wise form.
        `jobs`: circuits of 64-unit dense / CP-T / mixing / Hadamard layers (the reference's learning notebook, BASELINE config 4)
        step as level launches over jobs (cirkit_amd/train_jobs.py) when the fused form does not apply: None where the plan
        qualifies, True insists, False never.  `fuse_optimizer` (fused and job forms, one rank): `step` runs the optimizer inside the backward
        epilogues -- the workgroup that holds a weight's gradient updates its logits and moments and writes the next step's
        softmax; no gradient, no normalised weight and no optimizer launch for those tensors (`loss_and_grads` still leaves
        every gradient in `grads`)."""
        if plan.semiring != "lse-sum":
            raise NotImplementedError("HipTrainer trains circuits under the real lse-sum semiring; squared circuits compiled under "
                                      "complex-lse-sum (Embedding / CP-T for c, ConstantValue / Hadamard / TensorDot for Z) train with "
                                      "cirkit_amd.training_squared.HipSquaredTrainer")
        if optimizer not in ("adam", "sgd"):
            raise ValueError(f"unknown optimizer {optimizer!r}")
        self.user_plan, self._pad_info = plan, None
        if pad_units:
            from . import padding

            res = padding.pad_units(plan)
            if res is not None:
                plan, self._pad_info = res
                host = {n: (tensors[n].detach().cpu().numpy() if hasattr(tensors[n], "detach") else np.asarray(tensors[n]))
                        for n in self.user_plan.tensors}
                tensors = padding.pad_tensors(self._pad_info, host)
        # all parameters live in ONE flat buffer (the store's tensors are views of it, in plan order -- the order of
        # the flat gradient and moment buffers): the optimizer step is a single launch
        dev = torch.device(device)
        names = list(plan.tensors)
        sizes = [int(np.prod(plan.tensors[n][0])) for n in names]
        self._flat_param = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        if isinstance(tensors, TensorStore):
            src = {n: tensors[n] for n in names}
        else:
            src = tensors
        store = TensorStore(dev)
        off = 0
        for n, sz in zip(names, sizes):
            view = self._flat_param[off : off + sz].view(plan.tensors[n][0])
            v = src[n]
            view.copy_(torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach().to(torch.float32))
            store._t[n] = view
            off += sz
        store.version += 1
        self.plan = plan
        self.fused, self._fz = False, None
        self._fuse_optimizer = bool(fuse_optimizer)
        self._opt: torch.Tensor | None = None  # the DEVICE ck_opt_state of the fused form's optimizer epilogues
        self._opt_key = None
        why = "fused=False" if fused is False else self._setup_fused(plan, store, device)
        if why is not None:
            if fused is True:
                raise NotImplementedError(f"fused training does not apply to this plan: {why}")
            # layer-wise forward, every activation materialised, row-major linear weights
            self.circuit = HipCircuit(plan, store, device=device, use_graph=False, fuse=False,
                                      batch_params=True, tiled_weights=False, dense_on_table=False, pad_units=False,
                                      fused_weight_softmax=False)
        self.device = self.circuit.device
        self._jobs = None
        self.lr, self.optimizer, self.betas, self.eps = lr, optimizer, betas, eps
        self.step_count = 0
        self._clock: str | None = None  # which optimizer clock has advanced: "device" (fused job step) | "host" (apply_gradients)
        self._grads_current = False  # `grads` holds the gradients of the last step (false after a fused job step)
        c = self.circuit
        if len(c._out_pairs) != 1:
            raise NotImplementedError("training needs a single circuit output")
        if not self.fused:
            self._check_supported()
        # one flat gradient buffer; per-tensor gradients are views of it (single all-reduce)
        self._flat_grad = torch.zeros(sum(sizes), dtype=torch.float32, device=self.device)
        self.grads: dict[str, torch.Tensor] = {}
        off = 0
        for n, sz in zip(names, sizes):
            self.grads[n] = self._flat_grad[off : off + sz].view(plan.tensors[n][0])
            off += sz
        self._m1 = torch.zeros_like(self._flat_grad) if optimizer == "adam" else None
        self._m2 = torch.zeros_like(self._flat_grad) if optimizer == "adam" else None
        self._moments: dict[str, tuple] = {}
        if optimizer == "adam":
            off = 0
            for n, sz in zip(names, sizes):
                self._moments[n] = (self._m1[off : off + sz], self._m2[off : off + sz])
                off += sz
        self._bwd: dict[int, dict] = {}
        # input validation: the circuit's flag is raised by a batch with an out-of-range category; a step on such a batch
        # changes nothing (`step`), the flag is latched into `_bad_seen` -- what `check_inputs()` reports -- and cleared
        self._bad_seen = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._step_flag = torch.zeros(1, dtype=torch.int32, device=self.device)  # fused: the flag of the step being taken
        self._skipped = torch.zeros(1, dtype=torch.int32, device=self.device)  # Adam steps that did not count
        if not self.fused and jobs is not False and self._pad_info is None:
            from .train_jobs import JobStep

            js = JobStep(self)
            if js.why is None:
                self._jobs = js
            elif jobs is True:
                raise NotImplementedError(f"the job form of the training step does not apply to this plan: {js.why}")
        elif jobs is True:
            raise NotImplementedError("the job form of the training step needs an unpadded plan outside the fused form")

    # ------------------------------------------------------------------------------------------
    _PARAM_OPS = {"tensor", "softmax", "log_softmax", "sigmoid", "exp", "log", "square", "clamp", "softplus", "scaled_sigmoid", "mixing_weight", "matmul"}

    def _check_supported(self) -> None:
        for spec, l in zip(self.plan.layers, self.circuit.layers):
            if isinstance(l, HipCategoricalLayer):
                if l.probs is None or l.probs.softmax_source() is None:
                    raise NotImplementedError("training: Categorical layers need probs = softmax(tensor)")
            elif isinstance(l, HipGaussianLayer):
                if l.log_partition is not None or (set(l.mean.ops) | set(l.stddev.ops)) - self._PARAM_OPS:
                    raise NotImplementedError("training: Gaussian layers with a log-partition or exotic parameters")
            elif isinstance(l, (HipSumLayer, HipCPTLayer)) and type(l) in (HipSumLayer, HipCPTLayer, HipTuckerLayer):
                if set(l.weight.ops) - self._PARAM_OPS:
                    raise NotImplementedError(f"training: weight parameterisation {l.weight.ops}")
            elif isinstance(l, (HipHadamardLayer, HipKroneckerLayer)):
                pass
            else:
                raise NotImplementedError(f"training: layer type {spec.type!r}")

    def _fast_softmax(self, l) -> bool:
        """tensor -> softmax weights evaluated by the batched prologue: their backward is one kernel."""
        return l.weight.ops == ["tensor", "softmax"] and l.weight.softmax_source() is not None and not l._mixing

    # ---- the fused form ----------------------------------------------------------------------------------------------
    def _setup_fused(self, plan: Plan, store: TensorStore, device) -> str | None:
        """Build the fused training circuit; returns None on success, else why the plan does not qualify."""
        if self._pad_info is not None:
            return "padded unit counts"
        # (`cache_params`: with the optimizer in the backward epilogues -- `step`, one rank -- the launches that update the logits
        #  write the derived parameters of the next forward themselves; the prologue runs only after the store was changed
        #  from outside)
        c = HipCircuit(plan, store, device=device, use_graph=False, fuse=True, batch_params=True, tiled_weights=False,
                       dense_on_table=True, pad_units=False, fused_weight_softmax=False, persistent_leaf=True,
                       params_at_end=False, keep_levels=True, direct_input=True, cache_params=self._fuse_optimizer)
        if len(c._out_pairs) != 1 or c._signed or len(c._groups) != 1:
            return "needs one output and exactly one fused leaf region"
        g = c._groups[0]
        c._ensure_param_batch()
        cat = c.layers[g.input_layer]
        if g.depth not in (2, 4) or g.dense_layer is None or g.root not in c._table_fused or not c.linear_levels:
            return "the leaf region must be Categorical -> dense -> 2 or 4 CP-T levels with the table built by one prologue job"
        if not isinstance(cat, HipCategoricalLayer) or cat.num_output_units != 32 or cat.num_categories > 256:
            return "the leaf region needs a 32-unit Categorical input layer of at most 256 categories"
        covered = set(g.virtual) | {g.root} | set(c._tail)
        if covered != set(range(len(c.layers))) or c._tdense or c._cp_blocks or c._regions or c._input_prod:
            return "layers outside the leaf region and the tail"
        if c._tail and not c._tail16_ok():
            return "the tail does not fit the 16-row walk"
        for j in list(g.levels) + [g.dense_layer] + list(c._tail):
            l = c.layers[j]
            if not (isinstance(l, (HipSumLayer, HipCPTLayer)) and type(l) in (HipSumLayer, HipCPTLayer) and self._fast_softmax(l)
                    and l.num_input_units == 32):
                return f"layer {j}: weights must be softmax(tensor) over 32 inputs"
        for j in g.levels:
            if c.layers[j].num_output_units != 32 or c.layers[j].arity != 2:
                return "fused levels must be binary CP-T layers of 32 units"
        if cat.probs is None or cat.probs.softmax_source() is None:
            return "Categorical layers need probs = softmax(tensor)"
        leaf_of_dense = c._children[g.dense_layer][:, 0, 1].astype(np.int64)
        if not np.array_equal(leaf_of_dense, np.arange(cat.num_folds)):
            return "the dense layer must read the Categorical folds in order"
        Cn = cat.num_categories
        if (((Cn + 1 + 31) // 32) + 3) * 4096 + 8 * 4096 > 160 * 1024:
            return "too many categories for the table backward's LDS"
        # every parameter gradient of the fused backward is WRITTEN, exactly once, by the launch that owns its tensor (nothing
        # zeroes the flat gradient): the tensors behind the Categorical table, the dense layer, the levels and the tail must be
        # pairwise distinct and cover the plan's tensors
        owned = [cat.probs.graph.nodes[0].config["tensor"]] + [
            c.layers[j].weight.graph.nodes[0].config["tensor"] for j in [g.dense_layer] + list(g.levels) + list(c._tail)]
        if len(set(owned)) != len(owned) or set(owned) != set(plan.tensors):
            return "parameter tensors shared between layers (or not reached by any layer)"
        self.circuit, self.fused = c, True
        # wavefronts per workgroup of the backward walk: 8 = two per SIMD; 4 = one per SIMD with the next unit's tiles in
        # flight (ck_leaf_bwd.hip), within 3 %
        self._bwd_waves = int(os.environ.get("CK_BWD_WAVES", "8"))
        dev = c.device
        dl = c.layers[g.dense_layer]
        kl = 1 << g.depth
        nodes = np.asarray(g.nodes).astype(np.int64)
        n_roots = c.layers[g.root].num_folds
        off = [int(v) for v in g.node_off]
        var_of_leaf = cat.scope_idx[:, 0].astype(np.int64)

        def lvl(l: int, t: int, j: int) -> int:  # fold of the j-th node of level l under root t (level 0: table folds)
            return int(nodes[off[l] + t * (kl >> l) + j])

        launches = []  # top first: (unit table, level of P)
        for top in range(g.depth, 0, -2):
            per_root = kl >> top  # nodes of level `top` per root
            tab = np.zeros((n_roots * per_root, 16), dtype=np.int32)
            for t in range(n_roots):
                for j in range(per_root):
                    r = tab[t * per_root + j]
                    r[0] = lvl(top, t, j) if top == g.depth else lvl(top + 1, t, j >> 1)
                    r[1] = lvl(top, t, j)
                    r[2], r[3] = lvl(top - 1, t, 2 * j), lvl(top - 1, t, 2 * j + 1)
                    for i in range(4):
                        r[4 + i] = lvl(top - 2, t, 4 * j + i)
                        if top == 2:
                            r[8 + i] = var_of_leaf[int(nodes[g.leaf_off + t * kl + 4 * j + i])]
                    r[12] = t
            launches.append((torch.from_numpy(tab).to(dev), top))
        # Categorical scatter: table fold d takes the gradient tile of the level-1 node above it
        gfold = np.zeros(dl.num_folds, dtype=np.int32)
        var_of_table = np.zeros(dl.num_folds, dtype=np.int64)
        for t in range(n_roots):
            for i in range(kl):
                d = lvl(0, t, i)
                gfold[d] = lvl(1, t, i >> 1)
                var_of_table[d] = var_of_leaf[int(nodes[g.leaf_off + t * kl + i])]
        # the two leaves under a level-1 node read the same gradient tile: their workgroups are placed 8 apart (same XCD, same time)
        by_tile: dict[int, list[int]] = {}
        for d in range(dl.num_folds):
            by_tile.setdefault(int(gfold[d]), []).append(d)
        groups = list(by_tile.values())
        fold_order = []
        for i0 in range(0, len(groups), 8):  # 8 groups at a time: member m of group j -> block 8 m + j of this stretch
            chunk = groups[i0:i0 + 8]
            for m in range(max(len(gr) for gr in chunk)):
                fold_order += [gr[m] for gr in chunk if m < len(gr)]
        assert sorted(fold_order) == list(range(dl.num_folds))
        self._fz = {
            "group": g, "launches": launches, "fold_order": torch.from_numpy(np.asarray(fold_order, dtype=np.int32)).to(dev),
            "gfold": torch.from_numpy(gfold).to(dev), "var": torch.from_numpy(var_of_table).to(dev),
            "per_B": {},
        }
        return None

    def _fused_binding(self, B: int, bd) -> dict:
        fz = self._fz
        hit = fz["per_B"].get(B)
        if hit is not None and hit["arena_ptr"] == bd.arena.data_ptr():
            return hit
        from .fusion import balanced_segments

        c, g = self.circuit, fz["group"]
        dev = self.device
        n_tiles = (B + 31) // 32
        hit = {"arena_ptr": bd.arena.data_ptr(), "work": [], "G": []}
        for tab, top in fz["launches"]:
            hit["work"].append(torch.from_numpy(balanced_segments(int(tab.shape[0]), n_tiles, c._n_cu, waves=self._bwd_waves)).to(dev))
            # the tiles this launch leaves for the level below its Q nodes (one per Q node)
            # (tile-native between two of these launches, row-major where the Categorical scatter reads them)
            nq = c.layers[g.levels[top - 2]].num_folds
            hit["G"].append(torch.empty((nq, B, 32) if top == 2 else (nq, n_tiles, 1024), dtype=torch.float32, device=dev))
        while len(fz["per_B"]) >= 4:
            fz["per_B"].pop(next(iter(fz["per_B"])))
        fz["per_B"][B] = hit
        return hit

    def _tail_bwd_tables(self, B: int, bd, st: dict, fb: dict) -> dict | None:
        """Descriptors of `ck_tail_bwd` -- the few-fold layers above the leaf region in ONE backward launch -- for this
        binding, or None when a layer does not qualify (then they run layer by layer, `_bwd_sum_layer`): CP-T / arity-1
        layers of 32 input units and 32 outputs (1 for a scalar root), every child read by exactly one fold."""
        key = (bd.arena.data_ptr(), st["garena"].data_ptr())
        hit = fb.get("tail_bwd")
        if hit is not None and hit["key"] == key:
            return hit["tabs"]
        c = self.circuit
        tabs = None
        layers = list(reversed(c._tail))
        ok = bool(layers) and os.environ.get("CK_TAIL_BWD", "1") != "0"
        for i in layers:
            l = c.layers[i]
            ok = ok and (st["flags"][i] == 0 and st["shared"].get(i) is None and l.num_input_units == 32 and l.arity <= 2
                         and (l.num_output_units == 32 or (l.num_output_units == 1 and l.num_folds == 1 and i == layers[0]))
                         and (l._mode == capi.CK_SUM_PROD or l.arity == 1) and not l.is_complex and l._w_layout == capi.CK_W_ROWMAJOR)
        if ok:
            n_tiles = (B + 31) // 32
            dt = np.dtype([("w", "<u8"), ("gout", "<u8"), ("dw_part", "<u8"), ("child", "<u8", 4), ("gchild", "<u8", 4), ("H", "<i4"), ("Ko", "<i4")])
            assert dt.itemsize == 96
            n_folds = sum(c.layers[i].num_folds for i in layers)
            stride = sum(c.layers[i].num_folds * c.layers[i].num_output_units * 32 for i in layers)
            part = torch.empty(n_tiles * stride, dtype=torch.float32, device=self.device)
            tab = np.zeros(n_folds, dtype=dt)
            level_begin, k, off, part_of = [0], 0, 0, {}
            arena, garena = bd.arena.data_ptr(), st["garena"].data_ptr()
            for i in layers:
                l = c.layers[i]
                ro = bd.row_off[i].cpu().numpy().reshape(l.num_folds, l.arity)
                part_of[i] = part.data_ptr() + 4 * off
                wbytes = l.num_output_units * 32 * 4
                for f in range(l.num_folds):
                    r = tab[k]
                    r["w"] = l._w.data_ptr() + f * wbytes
                    r["gout"] = st["gviews"][i].data_ptr() + f * B * l.num_output_units * 4
                    r["dw_part"] = part_of[i] + f * wbytes
                    for h in range(2):  # (a single child is named twice: the launch issues a fixed number of loads and stores)
                        r["child"][h] = arena + 4 * int(ro[f, min(h, l.arity - 1)])
                        r["gchild"][h] = garena + 4 * int(ro[f, min(h, l.arity - 1)])
                    r["H"], r["Ko"] = l.arity, l.num_output_units
                    k += 1
                off += l.num_folds * l.num_output_units * 32
                level_begin.append(k)
            tabs = {"folds": torch.from_numpy(tab.view(np.uint8).reshape(n_folds, -1)).to(self.device), "n_folds": n_folds,
                    "levels": torch.from_numpy(np.asarray(level_begin, dtype=np.int32)).to(self.device), "n_levels": len(layers),
                    "part": part, "part_of": part_of, "stride": stride, "n_tiles": n_tiles}
        fb["tail_bwd"] = {"key": key, "tabs": tabs}
        fb.pop("sm_jobs", None)
        fb.pop("sm_jobs_opt", None)
        return tabs

    def _softmax_bwd_jobs(self, st: dict, fb: dict, tail: dict | None, with_opt: bool = False) -> tuple[torch.Tensor, int]:
        """(device job table, blocks) of `ck_param_softmax_bwd_batch` over every sum layer of the fused trainer; the tail
        layers' weight gradients are the per-tile slots `ck_tail_bwd` left (summed by that launch) when `tail` is given.
        `with_opt`: the jobs also name the logits, their moments and the evaluated weights (the optimizer epilogue)."""
        name_ = "sm_jobs_opt" if with_opt else "sm_jobs"
        hit = fb.get(name_)
        key = (st["dw_flat"].data_ptr(), None if tail is None else tail["part"].data_ptr())
        if hit is None or hit[0] != key:
            c, g = self.circuit, self._fz["group"]
            rows = []
            for j in list(c._tail) + list(g.levels):  # (the dense layer's is part of ck_table_dense_bwd)
                l = c.layers[j]
                w = l._w
                parted = tail is not None and j in tail["part_of"]
                name = l.weight.graph.nodes[0].config["tensor"]
                opt = (0, 0, 0, 0)
                if with_opt:
                    m1, m2 = self._moments.get(name, (None, None))
                    opt = (self.circuit.store[name].data_ptr(), 0 if m1 is None else m1.data_ptr(), 0 if m2 is None else m2.data_ptr(), w.data_ptr())
                rows.append((w.data_ptr(), tail["part_of"][j] if parted else st["dws"][j].data_ptr(),
                             self.grads[name].data_ptr(), l.num_folds * l.num_output_units,
                             l.num_input_units, tail["stride"] if parted else 0, tail["n_tiles"] if parted else 0, *opt))
            jt = np.zeros(len(rows), dtype=np.dtype([("w", "<u8"), ("dw", "<u8"), ("dtheta", "<u8"), ("rows", "<i8"), ("len", "<i4"), ("first", "<i4"),
                                                     ("part_stride", "<i8"), ("n_part", "<i4"), ("reserved", "<i4"),
                                                     ("theta", "<u8"), ("m1", "<u8"), ("m2", "<u8"), ("w_out", "<u8")]))
            assert jt.dtype.itemsize == 88
            first = 0
            for r, (w, dw, dt, n, ln, ps, npart, th, m1, m2, wo) in zip(jt, rows):
                r["w"], r["dw"], r["dtheta"], r["rows"], r["len"], r["first"], r["part_stride"], r["n_part"] = w, dw, dt, n, ln, first, ps, npart
                r["theta"], r["m1"], r["m2"], r["w_out"] = th, m1, m2, wo
                first += (n + 3) // 4
            hit = fb[name_] = (key, (torch.from_numpy(jt.view(np.uint8).reshape(len(rows), -1)).to(self.device), first))
        return hit[1]

    def _backward_fused(self, B: int, gB: float, seed, bd, st: dict, stream: int, with_opt: bool = False) -> None:
        c, fz = self.circuit, self._fz
        g = fz["group"]
        fb = self._fused_binding(B, bd)
        if fz.get("dw_sum_key") != st["dw_flat"].data_ptr():
            # the part of the flat linear-gradient buffer that float atomics add to: everything but the table gradient,
            # which the scatter overwrites (it is the last block: the Categorical layer comes first in the plan ... or not)
            dT = st["dws"][g.input_layer]
            flat = st["dw_flat"]
            lo = (dT.data_ptr() - flat.data_ptr()) // 4
            hi = lo + dT.numel()
            if lo == 0:
                fz["dw_sum"] = flat[hi:]
            elif hi == flat.numel():
                fz["dw_sum"] = flat[:lo]
            else:
                fz["dw_sum"] = flat
            fz["dw_sum_key"] = flat.data_ptr()
        keep, redo = bd.keep[g.root]
        gviews = st["gviews"]
        # ONE fill: the linear-space weight gradients (float atomics add to them).  The parameter gradients themselves are
        # written, each exactly once, by the parameter backward launches; the table gradient by the scatter.  The launch also
        # turns the validation flag of the forward into this step's flag (`step`: what the optimizer launch skips on)
        # (with the optimizer in the epilogues -- `with_opt`, one rank -- it is the optimizer's clock as well: a flagged step is dropped)
        capi.call("ck_fill_latch", fz["dw_sum"].data_ptr(), fz["dw_sum"].numel(), 0.0, c._bad_input.data_ptr(),
                  self._step_flag.data_ptr(), self._bad_seen.data_ptr(), self._opt_state().data_ptr() if with_opt else None, stream)
        for p in st["need_zero"]:
            if gviews[p] is not None:
                capi.call("ck_fill_f32", gviews[p].data_ptr(), gviews[p].numel(), 0.0, stream)
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        if c.layers[po].num_output_units != 1:
            raise NotImplementedError("training needs a scalar output unit")
        if gviews[po].numel() != B:
            capi.call("ck_fill_f32", gviews[po].data_ptr(), gviews[po].numel(), 0.0, stream)
        if seed is None:  # (nobody writes this block: the constant seed of the mean log-likelihood is filled once per binding)
            key = (gviews[po].data_ptr(), B, float(gB))
            if fb.get("seed_key") != key:
                capi.call("ck_fill_f32", gviews[po][fo].data_ptr(), B, -1.0 / gB, stream)
                fb["seed_key"] = key
        else:
            gviews[po][fo].reshape(-1)[:B].copy_(seed.reshape(-1))
            fb["seed_key"] = None
        tail = self._tail_bwd_tables(B, bd, st, fb)
        if tail is not None:  # the few-fold layers above the leaf region: one launch, a workgroup per 32-row tile
            capi.call("ck_tail_bwd", tail["folds"].data_ptr(), tail["n_folds"], tail["levels"].data_ptr(), tail["n_levels"], B,
                      tail["stride"], stream)
        else:
            for i in reversed(c._tail):  # ... or layer by layer
                self._bwd_sum_layer(i, bd, st, B, stream)
        # the leaf region, two levels per launch, top first
        cat, dl = c.layers[g.input_layer], c.layers[g.dense_layer]
        gin = gviews[g.root]
        for k, (tab, top) in enumerate(fz["launches"]):
            lp, lq = g.levels[top - 1], g.levels[top - 2]
            d = capi.LeafBwdLaunch()
            d.unit_tab, d.work = tab.data_ptr(), fb["work"][k].data_ptr()
            d.n_seg, d.n_wg, d.B, d.waves = int(fb["work"][k].shape[0]), c._n_cu, B, self._bwd_waves
            d.C, d.D, d.leaf = cat.num_categories, self.plan.num_variables, 1 if top == 2 else 0
            d.gin, d.gin_rowmajor = gin.data_ptr(), 1 if k == 0 else 0
            d.y_p = keep[top - 1].data_ptr()  # (the level in between, top - 1, is recomputed by the launch)
            if top == 2:
                d.table, d.x_rows = c._group_dev[g.root][1].data_ptr(), bd.x_last.data_ptr()
            else:
                d.y_c = keep[top - 3].data_ptr()
            d.w_p, d.w_q = c.layers[lp]._w.data_ptr(), c.layers[lq]._w.data_ptr()
            d.dw_p, d.dw_q = st["dws"][lp].data_ptr(), st["dws"][lq].data_ptr()
            d.gout = fb["G"][k].data_ptr()
            d.redo = redo.data_ptr()
            capi.call("ck_leaf_walk_bwd", C.byref(d), stream)
            gin = fb["G"][k]
        # (root, tile) units whose forward walk left the linear range: in log space, by a launch in which every other wave exits
        depth = g.depth
        capi.call("ck_leaf_walk_bwd_redo", c._group_dev[g.root][1].data_ptr(), c._group_dev[g.root][3].data_ptr(), bd.x_last.data_ptr(),
                  B, cat.num_categories, self.plan.num_variables, c._group_dev[g.root][0].data_ptr(),
                  (C.c_int32 * (depth + 1))(*g.node_off), g.leaf_off, cat._scope(self.device).data_ptr(), depth,
                  (C.c_void_p * depth)(*[c.layers[j]._w.data_ptr() for j in g.levels]),
                  (C.c_void_p * depth)(*[st["dws"][j].data_ptr() for j in g.levels]),
                  gviews[g.root].data_ptr(), gin.data_ptr(), redo.data_ptr(), c.layers[g.root].num_folds, None, 0, stream)
        # leaves: scatter by category into the gradient of the (F0, C + 1, 32) table T' = dense(log-table) ...
        Cn = cat.num_categories
        dTp = st["dws"][g.input_layer]
        capi.call("ck_transpose_i64_to_i32", bd.x_last.data_ptr(), bd.xt_i.data_ptr(), B, self.plan.num_variables, stream)
        capi.call("ck_categorical_bwd", gin.data_ptr(), fz["gfold"].data_ptr(), bd.xt_i.data_ptr(), fz["var"].data_ptr(),
                  dTp.data_ptr(), dl.num_folds, B, 32, Cn, 0, (fz["fold_order"].data_ptr() if B >= 256 else None), stream)
        # ... then the dense layer and the log-softmax of the Categorical layer backward ON THE TABLE (C + 1 rows per fold)
        n_cat, n_dense = cat.probs.graph.nodes[0].config["tensor"], dl.weight.graph.nodes[0].config["tensor"]
        topt, state = None, None
        if with_opt:
            # the optimizer in the epilogues (one rank): the launch that
            # holds the gradients of the Categorical and dense logits updates them and writes the next forward's table, the
            # launch that differentiates the weight softmaxes updates those logits and writes the next forward's weights
            state = self._opt_state().data_ptr()  # (its clock of this step: the fill launch at the start of the list)
            topt = capi.TableOpt()
            topt.state = state
            (m1c, m2c), (m1d, m2d) = self._moments.get(n_cat, (None, None)), self._moments.get(n_dense, (None, None))
            ph = self._flat_grad  # (SGD: the moment pointers are never read)
            topt.m1_cat, topt.m2_cat = (ph if m1c is None else m1c).data_ptr(), (ph if m2c is None else m2c).data_ptr()
            topt.m1_dense, topt.m2_dense = (ph if m1d is None else m1d).data_ptr(), (ph if m2d is None else m2d).data_ptr()
            topt.table, topt.table_scale = c._group_dev[g.root][1].data_ptr(), c._group_dev[g.root][3].data_ptr()
        capi.call("ck_table_dense_bwd", cat.probs.softmax_source().data_ptr(), None, dl.weight.softmax_source().data_ptr(),
                  dTp.data_ptr(), self.grads[n_cat].data_ptr(), self.grads[n_dense].data_ptr(), dl.num_folds, Cn,
                  None if topt is None else C.byref(topt), stream)
        # softmax parameterisation of every sum layer's weights (tail, fused levels, dense layer): one launch
        jobs, n_blocks = self._softmax_bwd_jobs(st, fb, tail, with_opt)
        capi.call("ck_param_softmax_bwd_batch", jobs.data_ptr(), jobs.shape[0], n_blocks, state, stream)

    def _accumulate_flags(self) -> tuple[dict[int, int], set[int]]:
        """Per consumer layer: 0 store / 1 add / 2 atomic; and the producer layers whose gradient
        block must be zeroed first.  A layer whose children -- (producer, fold) pairs -- are read by nobody else (any
        tree-structured region graph) STORES their gradients: nothing to zero, nothing to add."""
        c = self.circuit
        count: dict[tuple[int, int], int] = {}
        dup_in_layer: dict[int, bool] = {}
        for j, ch in enumerate(c._children):
            if ch is None:
                continue
            pairs = ch.reshape(-1, 2)
            uniq = np.unique(pairs, axis=0)
            dup_in_layer[j] = len(uniq) != len(pairs)
            for p, f in uniq:
                count[(int(p), int(f))] = count.get((int(p), int(f)), 0) + 1
        flags: dict[int, int] = {}
        for j, ch in enumerate(c._children):
            if ch is None:
                continue
            if dup_in_layer[j]:
                flags[j] = 2
            elif any(count[(int(p), int(f))] > 1 for p, f in ch.reshape(-1, 2)):
                flags[j] = 1
            else:
                flags[j] = 0
        # a launch with flag 1/2 adds into ALL its producers: they all must start from zero
        need_zero: set[int] = set()
        for j, fl in flags.items():
            if fl:
                need_zero |= {int(p) for p in np.unique(c._children[j][..., 0])}
        # (a storing layer may share a producer LAYER with an adding one: its own (producer, fold) blocks have no other writer
        #  -- that is what flag 0 means -- and the zero fills run before every backward launch, so the store loses nothing)
        return flags, need_zero

    def _bind_backward(self, B: int) -> dict:
        st = self._bwd.get(B)
        bd = self.circuit._bind(B)
        if st is not None and st["arena_ptr"] == bd.arena.data_ptr():
            return st
        c = self.circuit
        garena = torch.zeros_like(bd.arena)
        gviews = []
        base = 0
        for i, l in enumerate(c.layers):
            if bd.views[i] is None:  # (fused: never materialised)
                gviews.append(None)
                continue
            n = l.num_folds * B * l.num_output_units
            off = bd.views[i].data_ptr() - bd.arena.data_ptr()
            gviews.append(garena.view(torch.uint8)[off : off + 4 * n].view(torch.float32).view(l.num_folds, B, l.num_output_units))
        flags, need_zero = self._accumulate_flags()
        # linear-space weight gradients (dW, dTable) live in ONE flat buffer: a single fill per step
        sizes = {}
        for i, l in enumerate(c.layers):
            if isinstance(l, (HipSumLayer, HipCPTLayer)) and not (l.weight.ops == ["tensor"]):
                sizes[i] = tuple(l._w.shape) if l._w is not None else (l.num_folds, l.num_output_units, l.num_input_units)  # mixing layers: the (F, K, H) coefficients
            elif isinstance(l, HipCategoricalLayer):
                sizes[i] = (l.num_folds, l.num_categories + 1, l.num_output_units)
            elif isinstance(l, HipGaussianLayer):
                sizes[(i, "mean")] = sizes[(i, "stddev")] = (l.num_folds, l.num_output_units)
        flat = torch.zeros(sum(int(np.prod(sh)) for sh in sizes.values()) or 1, dtype=torch.float32, device=self.device)
        dws, off = {}, 0
        for i, sh in sizes.items():
            n = int(np.prod(sh))
            dws[i] = flat[off : off + n].view(sh)
            off += n
        # layers whose folds share children: contributions go to a temporary (one block per (fold, slot)) and are
        # added per distinct child by ck_segment_add_rows instead of through float atomics
        shared, tmp_elems = {}, 0
        for j, fl in flags.items():
            l = c.layers[j]
            if fl != 2 or bd.row_off[j] is None:
                continue
            ro = bd.row_off[j].cpu().numpy().reshape(-1)  # element offsets of the (fold, slot) children in the arena
            block = B * l.num_input_units
            uniq, inv = np.unique(ro, return_inverse=True)
            order = np.argsort(inv, kind="stable")
            cptr = np.concatenate([[0], np.cumsum(np.bincount(inv, minlength=len(uniq)))])
            dev = self.device
            shared[j] = {
                "row_off": (torch.arange(len(ro), dtype=torch.int64) * block).to(dev),
                "cptr": torch.from_numpy(cptr.astype(np.int32)).to(dev),
                "clist": torch.from_numpy(order.astype(np.int32)).to(dev),
                "coff": torch.from_numpy(uniq.astype(np.int64)).to(dev),
                "n_child": int(len(uniq)), "block": int(block),
            }
            tmp_elems = max(tmp_elems, len(ro) * block)
        tmp = torch.empty(max(tmp_elems, 1), dtype=torch.float32, device=self.device)
        st = {"arena_ptr": bd.arena.data_ptr(), "garena": garena, "gviews": gviews, "flags": flags,
              "need_zero": need_zero, "dws": dws, "dw_flat": flat, "shared": shared, "tmp": tmp}
        while len(self._bwd) >= 4:  # like the forward bindings: a handful of batch sizes stay resident
            self._bwd.pop(next(iter(self._bwd)))
        self._bwd[B] = st
        return st

    # ------------------------------------------------------------------------------------------
    def loss_and_grads(self, x: torch.Tensor, *, global_batch: int | None = None) -> torch.Tensor:
        """Forward + backward for ``loss = -(1/global_batch) sum_b log p(x_b)``; gradients land in
        ``self.grads`` (views of one flat buffer).  Returns the device tensor [sum log p, count] of this shard -- a view
        of the circuit's own buffer, overwritten by the next step (clone it to keep it).

        ``global_batch`` defaults to the number of rows of ALL ranks when torch.distributed is initialised (every rank is
        assumed to hold as many rows as this one; pass it explicitly otherwise), so that the SUM all-reduce of
        `all_reduce_grads` yields the gradient of the mean NLL of the global batch."""
        with torch.cuda.device(self.device):  # every launch below goes to a stream of self.device
            return self._loss_and_grads(x, global_batch)

    def _loss_and_grads(self, x: torch.Tensor, global_batch: int | None) -> torch.Tensor:
        self._grads_current = True
        import torch.distributed as dist

        if global_batch is None and _world_size() > 1:
            global_batch = int(x.shape[0]) * _world_size()
        B = int(x.shape[0])
        if self._jobs is not None:  # one recorded launch list: parameters, forward levels, root, backward levels
            return self._jobs.loss_and_grads(x, float(global_batch or B))
        ll = self._forward(x)
        self._backward(B, float(global_batch or B), None)
        return ll

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        """The training forward: [sum log p, count] of the batch; fused: kept tiles of the leaf region + tail outputs,
        layer-wise: every activation stays in the arena."""
        return self.circuit.log_likelihood_sum(x)

    def _backward(self, B: int, gB: float, seed: torch.Tensor | None, with_opt: bool = False) -> None:
        """The backward launch list over the activations of the LAST forward at batch size B: gradients of
        ``sum_b seed[b] * log p(x_b)`` (seed None: of ``-(1 / gB) sum_b log p(x_b)``) into `self.grads`."""
        c = self.circuit
        bd = c._bind(B)
        st = self._bind_backward(B)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.fused:
            return self._backward_fused(B, gB, seed, bd, st, stream, with_opt)
        capi.call("ck_fill_f32", st["dw_flat"].data_ptr(), st["dw_flat"].numel(), 0.0, stream)
        capi.call("ck_fill_f32", self._flat_grad.data_ptr(), self._flat_grad.numel(), 0.0, stream)
        gviews, flags = st["gviews"], st["flags"]
        for p in st["need_zero"]:
            capi.call("ck_fill_f32", gviews[p].data_ptr(), gviews[p].numel(), 0.0, stream)
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        if c.layers[po].num_output_units != 1:
            raise NotImplementedError("training needs a scalar output unit")
        capi.call("ck_fill_f32", gviews[po].data_ptr(), gviews[po].numel(), 0.0, stream)
        if seed is None:
            capi.call("ck_fill_f32", gviews[po][fo].data_ptr(), B, -1.0 / gB, stream)
        else:  # (an arbitrary gradient of the outputs: `HipCircuitModule` under autograd)
            gviews[po][fo].reshape(-1)[:B].copy_(seed.reshape(-1))
        shared, tmp = st["shared"], st["tmp"]

        def target(i):
            """(gradient arena, child offsets, accumulate flag) of layer i's backward launch."""
            sh = shared.get(i)
            if sh is None:
                return st["garena"].data_ptr(), bd.row_off[i].data_ptr(), flags[i]
            return tmp.data_ptr(), sh["row_off"].data_ptr(), 0

        def gather_shared(i):
            sh = shared.get(i)
            if sh is not None:
                capi.call("ck_segment_add_rows", tmp.data_ptr(), sh["cptr"].data_ptr(), sh["clist"].data_ptr(),
                          sh["coff"].data_ptr(), st["garena"].data_ptr(), sh["n_child"], sh["block"], stream)

        for i in range(len(c.layers) - 1, -1, -1):
            l = c.layers[i]
            if isinstance(l, HipCategoricalLayer):
                dT = st["dws"][i]
                capi.call("ck_categorical_bwd", gviews[i].data_ptr(), None, bd.xt_i.data_ptr(), l._scope(self.device).data_ptr(),
                          dT.data_ptr(), l.num_folds, B, l.num_output_units, l.num_categories, 1, None, stream)
                name = l.probs.graph.nodes[0].config["tensor"]
                capi.call("ck_param_log_table_bwd", l._table.data_ptr(), dT.data_ptr(), self.grads[name].data_ptr(),
                          l.num_folds, l.num_output_units, l.num_categories, 0, stream)
            elif isinstance(l, HipGaussianLayer):
                mean, stddev, _ = l._vals
                dm, ds = st["dws"][(i, "mean")], st["dws"][(i, "stddev")]
                capi.call("ck_gaussian_bwd", gviews[i].data_ptr(), bd.xt.data_ptr(), l._scope(self.device).data_ptr(),
                          mean.data_ptr(), stddev.data_ptr(), dm.data_ptr(), ds.data_ptr(), l.num_folds, B,
                          l.num_output_units, stream)
                l.mean.backward(dm, self.grads, stream)
                l.stddev.backward(ds, self.grads, stream)
            elif isinstance(l, HipHadamardLayer):
                ga, ro, fl = target(i)
                capi.call("ck_hadamard_bwd", ga, ro, gviews[i].data_ptr(), l.num_folds, l.arity, B, l.num_input_units, fl, stream)
                gather_shared(i)
            elif isinstance(l, HipKroneckerLayer):  # inner.py:178-187
                ga, ro, fl = target(i)
                capi.call("ck_kronecker_bwd", ga, ro, gviews[i].data_ptr(), l.num_folds, l.arity, B, l.num_input_units, fl, stream)
                gather_shared(i)
            elif l._mixing:
                dmw = st["dws"][i]
                ga, ro, fl = target(i)
                capi.call("ck_mixing_lse_bwd", bd.arena.data_ptr(), ga, bd.row_off[i].data_ptr(), ro,
                          l._w.data_ptr(), gviews[i].data_ptr(), dmw.data_ptr(), l.num_folds, l.arity, B,
                          l.num_output_units, fl, stream)
                gather_shared(i)
                if l._batched:  # tensor -> softmax evaluated by the batched prologue: its backward is one kernel
                    name = l.weight.graph.nodes[0].config["tensor"]
                    capi.call("ck_param_softmax_bwd", l._w.data_ptr(), dmw.data_ptr(), self.grads[name].data_ptr(),
                              l.num_folds * l.num_output_units, dmw.shape[-1], 0, stream)
                else:
                    l.weight.backward(dmw, self.grads, stream, upto=len(l.weight.graph.nodes) - 2)
            else:  # sum / cpt
                self._bwd_sum_layer(i, bd, st, B, stream)

    def _bwd_sum_layer(self, i: int, bd, st: dict, B: int, stream: int) -> None:
        """Backward launch of sum / CP-T layer i over its materialised inputs and output gradient, then its weight's
        parameter graph (semiring.py:383-408 under autograd)."""
        c = self.circuit
        l, gviews, flags = c.layers[i], st["gviews"], st["flags"]
        raw = l.weight.ops == ["tensor"]
        dW = self.grads[l.weight.graph.nodes[0].config["tensor"]] if raw else st["dws"][i]
        sh = st["shared"].get(i)
        if sh is None:
            ga, ro, fl = st["garena"].data_ptr(), bd.row_off[i].data_ptr(), flags[i]
        else:
            ga, ro, fl = st["tmp"].data_ptr(), sh["row_off"].data_ptr(), 0
        capi.call("ck_sum_lse_bwd", bd.arena.data_ptr(), ga, bd.row_off[i].data_ptr(), ro,
                  l._w.data_ptr(), bd.views[i].data_ptr(), gviews[i].data_ptr(), dW.data_ptr(), l.num_folds,
                  l.arity, B, l.num_input_units, l.num_output_units, l._mode, fl, stream)
        if sh is not None:
            capi.call("ck_segment_add_rows", st["tmp"].data_ptr(), sh["cptr"].data_ptr(), sh["clist"].data_ptr(),
                      sh["coff"].data_ptr(), st["garena"].data_ptr(), sh["n_child"], sh["block"], stream)
        if self.fused:
            return  # (one batched softmax backward for all layers at the end of `_backward_fused`)
        if self._fast_softmax(l):
            name = l.weight.graph.nodes[0].config["tensor"]
            rows = l.num_folds * l.num_output_units
            capi.call("ck_param_softmax_bwd", l._w.data_ptr(), dW.data_ptr(), self.grads[name].data_ptr(),
                      rows, dW.shape[-1], 0, stream)
        elif not raw:
            l.weight.backward(dW, self.grads, stream)

    def gradients(self) -> dict[str, np.ndarray]:
        """The gradients of the last `loss_and_grads`, host copies in the shapes of the user's plan."""
        if not self._grads_current:
            raise RuntimeError("gradients(): the last step was a fused job step, whose gradients never reach `grads` (the optimizer runs "
                               "in the job epilogues); call loss_and_grads() to obtain them")
        out = {}
        for n in self.user_plan.tensors:
            g = self.grads[n].detach().cpu().numpy()
            out[n] = self._pad_info.unpad(n, g) if self._pad_info is not None else g
        return out

    def parameters(self) -> dict[str, np.ndarray]:
        """The current parameter values, host copies in the shapes of the user's plan."""
        return {n: self.circuit.store.export(n) if self._pad_info is None else
                self._pad_info.unpad(n, self.circuit.store[n].detach().cpu().numpy()) for n in self.user_plan.tensors}

    def all_reduce_grads(self) -> None:
        """The one gradient exchange of data-parallel training: SUM over ranks of the flat buffer."""
        import torch.distributed as dist

        # (also at world size 1: the collective is then RCCL's identity, and the same call path is what a 1-GPU box can test)
        # RCCL through the C ABI (ck_comm_all_reduce_f32, on the launch stream) when a HipComm is set; torch.distributed otherwise
        if _default_comm() is not None or (dist.is_available() and dist.is_initialized()):
            _all_reduce_sum(self._flat_grad)

    def apply_gradients(self, skip_flag: torch.Tensor | None = None) -> None:
        """The optimizer step on `self.grads`.  `skip_flag`: a device int32; when it is nonzero at launch time the step
        changes nothing (parameters, moments, Adam's step count)."""
        with torch.cuda.device(self.device):
            self._apply_gradients(skip_flag)

    def _fused_opt_ok(self) -> bool:
        """The fused form takes the optimizer into its backward epilogues: `fuse_optimizer`, no padded duplicates to follow,
        a Categorical table the epilogue's table job applies to (C % 4 == 0; `_setup_fused` checked C <= 256)."""
        return (self.fused and self._fuse_optimizer and self._pad_info is None and self._jobs is None
                and self.circuit.layers[self._fz["group"].input_layer].num_categories % 4 == 0)

    def _opt_state(self) -> torch.Tensor:
        """The DEVICE `ck_opt_state` (constants, clock, dropped steps) of the optimizer epilogues."""
        key = (float(self.lr), tuple(float(b) for b in self.betas), float(self.eps))
        if self._opt is None:
            o = capi.OptState()
            o.lr, o.b1, o.b2, o.eps, o.bc1, o.bc2 = self.lr, self.betas[0], self.betas[1], self.eps, 1.0, 1.0
            o.step, o.skipped, o.skip_now, o.kind = 0, 0, 0, 1 if self.optimizer == "adam" else 0
            o.b1d, o.b2d = float(self.betas[0]), float(self.betas[1])
            self._opt = torch.frombuffer(bytearray(bytes(o)), dtype=torch.uint8).to(self.device)
        elif key != self._opt_key:  # (the learning rate was changed between steps: the first 16 bytes)
            head = torch.tensor([self.lr, self.betas[0], self.betas[1], self.eps], dtype=torch.float32).view(torch.uint8)
            self._opt[:16].copy_(head.to(self.device))
            self._opt[40:56].copy_(torch.tensor([self.betas[0], self.betas[1]], dtype=torch.float64).view(torch.uint8).to(self.device))
        self._opt_key = key
        return self._opt

    def _use_clock(self, which: str) -> None:
        """ONE optimizer clock per trainer: the fused job step counts Adam's steps on the device (`ck_opt_state.step`, not advanced
        by dropped batches), `apply_gradients` on the host (`step_count`).  Both update the same moments, so a trainer that has
        stepped with one refuses the other rather than applying inconsistent bias corrections."""
        if self._clock is not None and self._clock != which:
            raise RuntimeError(f"this trainer's optimizer clock is on the {self._clock}: step() (fused job form, one rank) and "
                               "loss_and_grads() + apply_gradients() (or a process group initialised mid-run) cannot be mixed on "
                               "one HipTrainer; construct it with fuse_optimizer=False to use the host clock throughout")
        self._clock = which

    def _apply_gradients(self, skip_flag: torch.Tensor | None = None) -> None:
        self._use_clock("host")
        self.step_count += 1
        stream = torch.cuda.current_stream(self.device).cuda_stream
        p, g = self._flat_param, self._flat_grad
        skip = None if skip_flag is None else skip_flag.data_ptr()
        if self.optimizer == "adam":
            capi.call("ck_adam_step", p.data_ptr(), g.data_ptr(), self._m1.data_ptr(), self._m2.data_ptr(), p.numel(),
                      self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, 1.0, skip, self._skipped.data_ptr(), stream)
        else:
            capi.call("ck_sgd_step", p.data_ptr(), g.data_ptr(), p.numel(), self.lr, 1.0, skip, stream)
        if self._pad_info is not None:  # padded input-layer units are copies of real ones: follow their update
            for n in self.user_plan.tensors:
                for ax, old, new in self._pad_info.duplicated_axes(n):
                    t = self.circuit.store[n]
                    idx = (torch.arange(old, new, device=t.device) % old)
                    t.narrow(ax, old, new - old).copy_(t.index_select(ax, idx))
        self.circuit.store.touch()  # values changed in place: circuits that cache derived parameters must refresh

    def step(self, x: torch.Tensor, *, global_batch: int | None = None) -> torch.Tensor:
        """One optimisation step on this rank's shard; returns the device tensor [sum log p, count]
        of the shard (before the update)."""
        import torch.distributed as dist

        c = self.circuit
        alone = _world_size() <= 1
        if self._jobs is not None and self._fuse_optimizer and alone:
            # the whole step is one recorded launch list; the optimizer runs where the gradients are (cirkit_amd/train_jobs.py)
            self._use_clock("device")
            self.step_count += 1
            self._grads_current = False
            return self._jobs.step(x, float(global_batch or int(x.shape[0])))
        if self._fused_opt_ok() and alone:
            # the fused form with the optimizer in its backward epilogues: no optimizer launch, no parameter prologue before
            # the next forward (`grads` still receives every gradient)
            self._use_clock("device")
            self.step_count += 1
            with torch.cuda.device(self.device):
                self._grads_current = True
                B = int(x.shape[0])
                ll = self._forward(x)
                self._backward(B, float(global_batch or B), None, with_opt=True)
            return ll
        ll = self.loss_and_grads(x, global_batch=global_batch)
        validate = c.validate_inputs and c._int_input
        # a batch with an out-of-range category (NaN log-likelihood) must not reach the parameters.  The flag it raised is THIS
        # step's (fused: handed on by the backward's first launch; layer-wise: latched and cleared below), everything stays on
        # the device -- no host synchronisation: on a single rank the optimizer launch changes nothing at all; with several
        # ranks the other ranks' gradients are valid and every rank must take the same step, so this rank's are dropped
        flag = self._step_flag if self.fused else c._bad_input
        if validate and not alone:
            with torch.cuda.device(self.device):
                capi.call("ck_zero_if_flag", self._flat_grad.data_ptr(), self._flat_grad.numel(), flag.data_ptr(),
                          torch.cuda.current_stream(self.device).cuda_stream)
        self.all_reduce_grads()
        self.apply_gradients(flag if (validate and alone) else None)
        if validate and not self.fused:
            with torch.cuda.device(self.device):
                capi.call("ck_latch_flag", c._bad_input.data_ptr(), self._bad_seen.data_ptr(),
                          torch.cuda.current_stream(self.device).cuda_stream)
        return ll

    @property
    def skipped_steps(self) -> int:
        """Steps that changed nothing because their batch held an illegal category (a device read)."""
        n = int(self._skipped.item())
        if self._jobs is not None:
            n += self._jobs.opt_counters()[1]
        if self._opt is not None:
            n += int(self._opt[28:32].cpu().view(torch.int32)[0])
        return n

    def check_inputs(self) -> None:
        """Raise ``IndexError`` if a batch since the last check held a category out of range (as
        `TorchCategoricalLayer`'s indexing would have, layers/input.py:399-412).  On a single rank the steps on such
        batches changed nothing (parameters, moments, Adam's step count); later valid batches train normally."""
        if int(self._bad_seen.item()) != 0:
            self._bad_seen.zero_()
            self.circuit._bad_input.zero_()
            raise IndexError("a batch held a category outside [0, num_categories) of its variable")
        self.circuit.check_inputs()


class _CircuitFunction(torch.autograd.Function):
    """``y = circuit(x)`` with the hand-written backward of `HipTrainer` behind it."""

    @staticmethod
    def forward(ctx, module, x, *params):
        tr = module._trainer
        with torch.cuda.device(tr.device):
            tr._forward(x)  # the training forward (what the backward needs stays on the device)
            B = int(x.shape[0])
            bd = tr.circuit._bind(B)
            po, fo = int(tr.circuit._out_pairs[0, 0]), int(tr.circuit._out_pairs[0, 1])
            y = bd.views[po][fo].reshape(B, 1, 1).clone()
        module._generation += 1
        ctx.module, ctx.B, ctx.generation = module, B, module._generation
        return y

    @staticmethod
    def backward(ctx, gout):
        m = ctx.module
        if ctx.generation != m._generation:
            raise RuntimeError("HipCircuitModule: backward of a forward that is not the last one (the activations live in ONE "
                               "arena: call backward before the next forward)")
        tr = m._trainer
        with torch.cuda.device(tr.device):
            tr._backward(ctx.B, float(ctx.B), gout.to(torch.float32).contiguous())
        return (None, None, *[tr.grads[n].clone() for n in m._names])


class HipCircuitModule(torch.nn.Module):
    """A ``torch.nn.Module`` over a plan: the reference's training loop, unchanged, on the plan-level HIP path --

        m = HipCircuitModule(plan, tensors);  opt = torch.optim.Adam(m.parameters(), lr=0.01)
        loss = -m(batch).mean();  loss.backward();  opt.step()          # notebooks/learning-a-circuit.ipynb, cell 18

    `forward` is the layer-wise HIP forward (``(B, 1, 1)`` log-likelihoods like ``TorchCircuit.forward``), `backward` the
    launch list of cirkit_amd/csrc/ck_backward.hip (`HipTrainer`), for ANY gradient of the outputs.  The parameters are
    ``nn.Parameter``s over the trainer's own storage (one flat buffer), so any torch optimizer updates the circuit in place.
    One forward at a time: its activations live in one arena, `backward` must run before the next `forward`.  Same
    coverage as `HipTrainer` (real lse-sum circuits with one scalar output)."""

    def __init__(self, plan: Plan, tensors: Mapping[str, object], *, device: str | torch.device = "cuda:0") -> None:
        super().__init__()
        self._trainer = HipTrainer(plan, tensors, device=device, optimizer="sgd", lr=0.0, jobs=False)  # (any output gradient)
        self._names = list(self._trainer.plan.tensors)
        self._generation = 0
        self.params = torch.nn.ParameterList([torch.nn.Parameter(self._trainer.circuit.store[n]) for n in self._names])

    def named_tensors(self) -> dict[str, torch.nn.Parameter]:
        return dict(zip(self._names, self.params))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _CircuitFunction.apply(self, x, *self.params)
