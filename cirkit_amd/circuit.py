"""`HipCircuit`: the MI355X-native counterpart of ``TorchCircuit`` for the forward pass.

Replaces, for one compiled circuit (reference lines in parentheses):

* ``TorchCircuit.forward`` / ``_evaluate_layers``                 circuits.py:242-278
* the interpreter loop ``TorchDiAcyclicGraph.evaluate``           graph/modules.py:303-335
* the gather of ``LayerAddressBook.lookup`` (cat + index copy)    circuits.py:30-71

Design (MI355X-first): all layer outputs live in ONE activation arena in HBM, laid
out ``(F, B, K)`` per layer; a fold index becomes a table of arena offsets that the kernels read
directly, so the reference's ``(F, H, B, K)`` gather copies (30 % of its CPU time, 2x the traffic
on a GPU) never exist.  For each batch size the launch sequence is recorded once into a native
``ck_program`` and replayed with one host call (optionally as a hipGraph), instead of a Python loop
with ~10 ATen launches per layer.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Mapping

import numpy as np
import torch

from . import _capi as capi
from . import padding
from .fusion import (CPBlock, RegionBlock, SubtreeGroup, find_cp_blocks, find_input_products, find_region_blocks,
                     find_subtree_groups, find_table_dense, find_tail)
from .layers import HipConstantValueLayer, HipEmbeddingLayer, HipInputLayer, HipLayer, HipTensorDotLayer, layer_from_spec
from .parameters import ParamBatch, TensorStore
from .circuit_launch import _LaunchMixin
from .circuit_profile import _ProfilingMixin
from .plan import Plan, resolve_fold_index

_ALIGN = 64  # arena alignment of every layer block, in activation elements (>= 256 B)


class _Binding:
    """Everything that depends on the batch size: arena, offset tables, recorded program."""

    def __init__(self) -> None:
        self.B = 0
        self.arena: torch.Tensor | None = None
        self.views: list[torch.Tensor] = []
        self.row_off: list[torch.Tensor | None] = []
        self.xt: torch.Tensor | None = None  # (D, B) fp32 staging copy of the batch
        self.xt_i: torch.Tensor | None = None  # (D, B) int32 staging copy
        self.leftover: dict[int, tuple[torch.Tensor, torch.Tensor]] = {}
        self.cp_tabs: dict[int, torch.Tensor] = {}  # device-address tables of the weight matrices (keep alive)
        self.program = None  # ck_program*
        self.program_ll = None  # the same followed by ck_ll_sum
        self.store_version = -1
        self.ll: torch.Tensor | None = None
        self.ll_cell = 0  # program input cell of program_ll that redirects the [sum, count] pair (0: none)
        self.params_at_end = False  # the tail launch evaluates the parameters of the next forward (ck_tail_params_fwd)
        self.keep: dict[int, tuple] = {}  # `keep_levels`: leaf group root -> ([(F_l, B, 32) linear tiles per level], tile flags)
        self.direct = False  # the leaf launches read the caller's int64 batch themselves (no staged copy of it)
        self.x_last: torch.Tensor | None = None  # ... the batch of the last call (kept alive; read by eager launches)

    def destroy(self) -> None:
        for name in ("program", "program_ll"):
            if getattr(self, name) is not None:
                capi.load().ck_program_destroy(getattr(self, name))
                setattr(self, name, None)


class HipCircuit(_LaunchMixin, _ProfilingMixin):
    """Evaluate a folded plan on one MI355X.

    Args:
        plan: the folded layer list (`cirkit_amd.plan.Plan`).
        tensors: parameter values by plan tensor name (numpy arrays or torch tensors).
        device: a ROCm device, e.g. ``"cuda:0"``.
        use_graph: replay each batch size's launch list as a hipGraph.  A graph launch leaves the GPU idle for ~8 us
            before its first kernel (measured, profiles/), a launch list of a handful of kernels replayed eagerly by
            the native executor none as long as the host stays ahead (measured on 4- to 30-kernel forwards at batch
            16 .. 4096: eager is 0.1 - 4 % faster); so graphs are only used for programs of more than
            `graph_min_launches` launches (default 64), where the host-side enqueue could become the bottleneck.
            False: never a graph.
        fuse: cross-layer fusion of the leaf region (cirkit_amd/fusion.py); an int caps the number
            of fused CP-T levels, False evaluates layer by layer (every layer output materialised).
        batch_params: recompute all softmax parameters with one launch per forward
            (`ck_param_softmax_batch`) instead of one launch per parameter node.
        contraction: how the K = 32 sum layers contract in linear space.  ``"f32"``: exact fp32 (v_mfma_f32_32x32x2_f32) -- the
            product, what every reported number uses.  ``"bf16x3"`` / ``"bf16x6"``: labelled VARIANTS of the depth-4 persistent
            leaf launch: every fp32 operand cut into two / three bf16 pieces (truncation, exact residuals; bf16 keeps fp32's
            exponent range), 3 / 6 products per contraction on the bf16 matrix pipe with fp32 accumulation -- ~2^-15 per
            product, resp. fp32-like (tests/test_gpu_parity.py measures both against the fp64 goldens) -- and of the stream-K
            launch of Tucker layers with 32 / 64 units (`ck_tucker_fwd`: the weights are cut into pieces while they are
            staged, behind the online softmax's exponential), and of the DMA-staged region / CP-block launches
            (`ck_region_lse_fwd_v`, `ck_cp_lse_fwd_v`: a weight unit is cut in LDS by the workgroup), and of the layer-wise
            dense / CP-T launches with 64..512 contracted inputs (`ck_sum_lse_fwd_v`).  The tail, the parameter jobs and
            every other launch stay exact fp32.
        dense_on_table: a Categorical input layer followed fold-by-fold by a dense sum layer only
            takes C distinct values per fold, so the dense layer is applied once per forward to the
            (F, C, K) log-probability table (same kernel, batch = C) instead of to every batch row;
            bit-identical results, B/C times less work for that layer.
        fused_weight_softmax: Tucker layers (arity 2, 32 / 64 units) whose weight is softmax(theta): the prologue skips them
            and the layer's stream-K launch reads the logits, normalising them online (`ck_tucker_logits_fwd`); the
            normalised (F, Ko, Ki^2) weights -- 1.6 GB at the reference's notebook configuration -- are never written nor
            read back.  Off for training (the backward kernels read the normalised weights).
        linear_levels: inside the fused leaf launch a value is handed from one CP-T level to the next as
            (linear tile, per-row log scale) instead of taking its log and exponentiating it again; the
            same sums with one log per row instead of 64 transcendentals (cirkit_amd/csrc/ck_fused.hip).
            False keeps the log / exp of the reference between the levels.
        cache_params: the reference re-evaluates every parameter graph (softmax, log, ...) on every
            forward and so does the default here.  True keeps the derived parameters of the last
            forward and recomputes them only after a parameter value changed (`TensorStore.set`,
            `invalidate_parameters`) -- the serving configuration.
        persistent_leaf: the fused leaf launch as ONE resident workgroup per CU walking (root, tile range) segments
            (cirkit_amd/csrc/ck_leaf.hip) instead of one workgroup per 128 rows.  None: whenever the launch is eligible
            (linear table, tiled fp32 weights) and has at least one 32-row tile per CU; bit-identical either way.
        validate_inputs: discrete inputs are range-checked on the device while they are staged (no extra launch, no host
            synchronisation): a category >= the layer's number of categories -- an ``IndexError`` in the reference -- makes
            the outputs NaN and `check_inputs()` raise.  Negative values are this library's "marginalised" sentinel.
        direct_input: when the persistent leaf launches are the only readers of a discrete batch, they read -- and validate --
            the caller's ``(B, D)`` int64 tensor themselves (`ck_leaf_walk_fwd` with a program input): no staging launch,
            no staged copy.  An out-of-range category then makes the outputs of ITS ROW NaN (and `check_inputs()` raise)
            instead of the whole batch's, and later batches are unaffected.  False always stages the batch.
        params_at_end: the parameter graphs are re-evaluated once per forward as in the reference, but at the END of a
            forward and for the next one: the launch that walks the tail carries the prologue's workgroups beside its own
            (`ck_tail_params_fwd`: both are latency-bound and independent, a tail block and a parameter block share a compute
            unit).  A forward whose `TensorStore` has changed since (`store.set`, `invalidate_parameters`) evaluates them at
            its start first (`TensorStore.state()`: `store.set` / `touch` and the torch version counters of the stored tensors,
            so an optimizer's in-place step is seen; only a write through a raw pointer by a foreign kernel needs `touch()`).
            Applies to circuits with ONE persistent leaf launch (table and dense layer built by one prologue job, C <= 256,
            tiled fp32 weights) and a 16-row tail: 27 us for the launch against 16.7 + 17.9 us for the two it replaces at
            the north-star configuration.  (Two further fusions -- the tail and the parameters INSIDE the leaf launch -- were
            built, were bit-identical and slower; LAB_NOTES.md has the measurements, the code is gone.)
        keep_levels: the training forward (cirkit_amd/training.py): every persistent leaf launch also stores the linear tile of
            each node it evaluates (`ck_leaf_walk_fwd` with keep_levels) -- what the fused backward (`ck_leaf_walk_bwd`) reads
            instead of the materialised layer outputs the reference's autograd keeps.  Needs `direct_input`.
        keep_layer_outputs: False: `forward` does not store the 32-unit fold outputs of the tail that only the tail itself
            reads (nobody but `layer_outputs()` wants them; `log_likelihood_sum` never stores them).
    """

    def __init__(
        self,
        plan: Plan,
        tensors: Mapping[str, object] | TensorStore,
        *,
        device: str | torch.device = "cuda:0",
        use_graph: bool = True,
        graph_min_launches: int = 64,
        signed_real: bool = True,
        fused_weight_softmax: bool = True,
        fuse: bool | int = True,
        batch_params: bool = True,
        contraction: str = "f32",
        dense_on_table: bool = True,
        tiled_weights: bool = True,
        cache_params: bool = False,
        fuse_regions: bool = True,
        linear_levels: bool = True,
        pad_units: bool = True,
        persistent_leaf: bool | None = None,
        validate_inputs: bool = True,
        direct_input: bool = True,
        keep_layer_outputs: bool = True,
        params_at_end: bool = True,
        keep_levels: bool = False,
        complex_linear: bool = True,
    ) -> None:
        if plan.semiring not in ("lse-sum", "complex-lse-sum"):
            raise ValueError(f"semiring {plan.semiring!r} is not evaluated by the HIP backend")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise capi.HipExtensionError("HipCircuit needs a ROCm device; there is no CPU fallback")
        capi.load()
        # unit counts that are not multiples of 32 are padded (cirkit_amd/padding.py): same function, MFMA tiles
        self.user_plan = plan
        self._pad_info = None
        if pad_units:
            if isinstance(tensors, TensorStore):
                hit = tensors._padded.get(id(plan))
                if hit is not None:
                    _, plan, self._pad_info = hit
                elif tensors._pad is not None:
                    # a store that another circuit padded: this plan must be padded the same way to read it
                    res = padding.pad_units(plan, tensors._pad.multiple)
                    ok = res is not None and all(
                        tuple(tensors[k].shape) == tuple(sh) for k, (sh, _) in res[0].tensors.items() if k in tensors)
                    if not ok and any(k in tensors and tuple(tensors[k].shape) != tuple(sh) for k, (sh, _) in plan.tensors.items()):
                        raise ValueError(
                            "this TensorStore holds parameters padded to multiples of 32 units by another HipCircuit, "
                            "and this plan cannot be padded the same way; build the first circuit with pad_units=False "
                            "to share its parameters")
                    if ok:
                        plan, self._pad_info = res
                        tensors._padded[id(self.user_plan)] = (self.user_plan, plan, self._pad_info)
            else:
                res = padding.pad_units(plan)
                if res is not None:
                    plan, self._pad_info = res
                    tensors = padding.pad_tensors(self._pad_info, tensors)
        self.plan = plan
        self.use_graph = use_graph
        self.graph_min_launches = int(graph_min_launches)
        self.cache_params = bool(cache_params)
        self.linear_levels = bool(linear_levels)
        self.persistent_leaf = persistent_leaf
        self.validate_inputs = bool(validate_inputs)
        self.direct_input = bool(direct_input)
        self.keep_layer_outputs = bool(keep_layer_outputs)
        self.params_at_end = bool(params_at_end)
        self.keep_levels = bool(keep_levels)
        self._params_valid_version = None  # store.state() the derived parameters in memory were evaluated from
        self._tailp: dict | None = None  # what the tail + parameters launch evaluates (`_plan_tail_params`)
        self._recording = False
        self._num_states: torch.Tensor | None = None
        self._states_consistent = True
        self._bad_input = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._n_cu = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        self._pprog = None
        self._pprog_version, self._pprog_data_version = -1, None
        if isinstance(tensors, TensorStore):
            self.store = tensors
        else:
            self.store = TensorStore(self.device)
            self.store.update(tensors)
            if self._pad_info is not None:
                self.store._pad = self._pad_info
                self.store._padded[id(self.user_plan)] = (self.user_plan, plan, self._pad_info)
        missing = [k for k in plan.tensors if k not in self.store]
        if missing:
            raise ValueError(f"missing parameter tensors: {missing}")
        self.layers: list[HipLayer] = [layer_from_spec(s, self.store, plan.semiring) for s in plan.layers]
        for l in self.layers:  # Tucker weights softmax(theta): the launch reads the logits (ck_tucker_logits_fwd)
            if hasattr(l, "_logits_ok"):
                l._logits_ok = bool(fused_weight_softmax)
            if hasattr(l, "_contraction"):  # (sum layers: the launches that have a bf16-piece variant)
                l._contraction = {"f32": 0, "bf16x3": 3, "bf16x6": 6}[contraction]
        self._folds = [l.num_folds for l in self.layers]
        self._complex = plan.semiring == "complex-lse-sum"
        self._act_dtype = torch.complex64 if self._complex else torch.float32
        # (producer, fold) pairs of every inner layer's children and of the circuit output
        self._children = [
            None if s.inputs is None else resolve_fold_index(s.inputs, self._folds) for s in plan.layers
        ]
        for s, ch, l in zip(plan.layers, self._children, self.layers):
            if ch is not None and ch.shape[:2] != (l.num_folds, l.arity):
                raise ValueError(
                    f"fold index of a {s.type} layer has shape {ch.shape[:2]}, expected {(l.num_folds, l.arity)}"
                )
        self._out_pairs = resolve_fold_index(plan.output, self._folds).reshape(-1, 2)
        data_inputs = [l for l in self.layers if isinstance(l, HipInputLayer) and not isinstance(l, HipConstantValueLayer)]
        self._float_input = any(l.wants_float_input for l in data_inputs)
        self._int_input = any(not l.wants_float_input for l in data_inputs)
        self._bindings: dict[int, _Binding] = {}
        self._scratch_buf: torch.Tensor | bool | None = None
        self._side: torch.cuda.Stream | None = None  # graphs cannot be captured on the null stream
        depth = 0 if fuse is False else (4 if fuse is True else int(fuse))
        # A complex-lse-sum circuit whose parameters are all real -- Embedding inputs and plain real sum weights, the
        # squared circuits of BASELINE config 5 -- is REAL-valued: the reference carries (log|v|, 0 or pi).  Its fused
        # launches then work on signed linear tiles (ck_leaf.hip, ck_tail16.hip with signed_values) instead of pairs of
        # complex exponentials; memory blocks stay complex64 as the reference's layer outputs are.
        self._signed = bool(signed_real) and self._complex and fuse is not False and persistent_leaf is not False \
            and linear_levels and self._is_real_valued()
        # A complex-lse-sum circuit with COMPLEX values (complex parameters, or `signed_real=False`) over one Embedding layer:
        # 32-unit CP-T / dense layers chained on linear (re, im) tiles, the complex logarithm taken once (circuit_clin.py,
        # csrc/ck_clin.hip).  It owns the whole launch list: none of the fusions below applies beside it.
        self._clin = None
        if self._complex and not self._signed and fuse is not False and complex_linear and linear_levels:
            from .circuit_clin import ClinPath

            self._clin = ClinPath.build(self, depth)
        if self._clin is not None:
            fuse = False
        self._groups: list[SubtreeGroup] = (
            find_subtree_groups(plan, self.layers, self._children, self._out_pairs, depth, signed=self._signed)
            if fuse is not False else []
        )
        if self._signed and (not self._groups or any(g.depth < 1 or self.layers[g.input_layer].num_states >= 65535 for g in self._groups)):
            self._signed, self._groups = False, []
        self._group_of_root = {g.root: g for g in self._groups}
        self._virtual = {i for g in self._groups for i in g.virtual}
        if self._clin is not None:
            self._virtual = self._clin.virtual_layers()
        self._group_dev: dict[int, tuple] = {}
        self._tail: list[int] = (
            find_tail(plan, self.layers, self._virtual | set(self._group_of_root), signed=self._signed) if fuse is not False else []
        )
        # dense sum layers evaluated inside the Hadamard layer that multiplies them (ck_cp.hip)
        self._table_fused: set[int] = set()  # group roots whose table + dense layer are one prologue job
        self._cp_blocks: dict[int, CPBlock] = {}
        self._cp_leftover: dict[int, np.ndarray] = {}
        self._cp_subset: dict[int, np.ndarray] = {}  # CP-block layers of which only these folds are evaluated
        self._regions: dict[int, RegionBlock] = {}
        self._input_prod: dict[int, int] = {}  # Hadamard layer -> the Gaussian layer it multiplies (ck_input.hip)
        self._input_prod_dev: dict[int, torch.Tensor] = {}
        # complex CP-T / dense layers (K = 32, real weights) whose children are all folds of ONE Embedding layer
        # that nobody else reads: they gather from its weight table, the Embedding output is never written
        self._emb_gather: dict[int, int] = {}
        self._emb_gather_dev: dict[int, tuple] = {}
        # (the gather launch reads a REAL weight table: circuits with a complex parameter anywhere evaluate the Embedding layer)
        if fuse is not False and self._complex and not any(self.store[n].is_complex() for n in self.store.names()):
            readers: dict[int, set[int]] = {}
            for j, ch in enumerate(self._children):
                if ch is not None:
                    for p in np.unique(ch[..., 0]):
                        readers.setdefault(int(p), set()).add(j)
            outs = {int(p) for p in self._out_pairs[:, 0]}
            for j, (sp, l) in enumerate(zip(plan.layers, self.layers)):
                ch = self._children[j]
                if ch is None or sp.type not in ("cpt", "sum") or (sp.type == "sum" and l.arity != 1):
                    continue
                if j in self._virtual or j in self._group_of_root or j in self._tail:
                    continue
                if l.num_input_units != 32 or l.num_output_units != 32 or l.weight.ops != ["tensor"]:
                    continue
                prods = np.unique(ch[..., 0])
                e = int(prods[0])
                if (len(prods) == 1 and isinstance(self.layers[e], HipEmbeddingLayer) and readers.get(e) == {j}
                        and e not in outs and self.layers[e].scope_idx.shape[1] == 1):
                    self._emb_gather[j] = e
                    self._virtual.add(e)
        # TensorDot layers of a product circuit with real parameters (the partition function of a squared circuit, BASELINE
        # config 5): the Hadamard layer beneath one is read as a list (never materialised), the W / conj W pair of a squared sum
        # layer is one launch (cirkit_amd/fusion.py tensordot_lists; 32 units everywhere: td32_fwd_kernel)
        self._td_had: dict[int, int] = {}
        self._td_pair: dict[int, int] = {}
        if fuse is not False and any(isinstance(l, HipTensorDotLayer) for l in self.layers) \
                and not any(self.store[n].is_complex() for n in self.store.names()):
            from .fusion import tensordot_lists

            busy = self._virtual | set(self._group_of_root) | set(self._tail) | set(self._emb_gather)
            self._td_had, self._td_pair = tensordot_lists(self.layers, self._children, {int(p) for p in self._out_pairs[:, 0]}, busy)
            self._virtual |= set(self._td_had.values())
        self._td_first = set(self._td_pair.values())  # (launched together with the layer above them)
        self._tdense: dict[int, int] = {}  # dense layer -> the Categorical layer it is tabulated over
        self._tdense_dev: dict[int, tuple] = {}  # dense layer -> (T' (F, C+1, 32), scope (F) int64, variables (F) numpy)
        if fuse is not False and dense_on_table and batch_params:
            busy = self._virtual | set(self._group_of_root) | set(self._tail)
            cand = find_table_dense(plan, self.layers, self._children, busy)
            def fits(d: int, c: int) -> bool:  # one fold's (K, C) block + statistics + weights in LDS
                K, Cn = self.layers[d].num_output_units, self.layers[c].num_categories
                return (K * (Cn + 1) + 2 * K + K * K) * 4 <= 160 * 1024

            self._tdense = {d: c for d, c in cand.items()
                            if self.layers[c].probs is not None and self.layers[c].probs.softmax_source() is not None
                            and self.layers[d].weight.softmax_source() is not None and fits(d, c)}
            # the block kernels (ck_cp.hip) take ONE category count for all the gather slots of a launch: dense layers
            # tabulated over Categorical layers with another number of categories than the most common one are simply
            # not tabulated (they are evaluated as ordinary dense folds)
            counts: dict[int, int] = {}
            for d, c in self._tdense.items():
                counts[self.layers[c].num_categories] = counts.get(self.layers[c].num_categories, 0) + self.layers[d].num_folds
            if len(counts) > 1:
                keep = max(counts, key=lambda k: (counts[k], k))
                self._tdense = {d: c for d, c in self._tdense.items() if self.layers[c].num_categories == keep}
            readers: dict[int, set[int]] = {}
            for j, ch in enumerate(self._children):
                if ch is not None:
                    for p in np.unique(ch[..., 0]):
                        readers.setdefault(int(p), set()).add(j)
            outs = {int(p) for p in self._out_pairs[:, 0]}
            for c in set(self._tdense.values()):  # a Categorical layer only read through tables is never evaluated
                if c not in outs and readers.get(c, set()) <= set(self._tdense):
                    self._virtual.add(c)
        if fuse is not False:
            self._input_prod = find_input_products(
                plan, self.layers, self._children, self._out_pairs, self._virtual | set(self._group_of_root) | set(self._tail))
            self._virtual |= set(self._input_prod.values())
            blocks, leftover, virt = find_cp_blocks(
                plan, self.layers, self._children, self._out_pairs,
                self._virtual | set(self._group_of_root) | set(self._tail) | set(self._input_prod))
            self._cp_blocks = {b.layer: b for b in blocks}
            self._cp_leftover = leftover
            self._virtual |= virt
            # mixing layers that take over the CP blocks they combine (ck_cp.hip: region_lse_kernel)
            regions, absorbed = find_region_blocks(
                plan, self.layers, self._children, self._out_pairs, self._cp_blocks,
                self._virtual | set(self._group_of_root) | set(self._tail) | set(self._input_prod)) if fuse_regions else ([], {})
            self._regions = {r.layer: r for r in regions}
            for h, mask in absorbed.items():
                if mask.all():
                    self._virtual.add(h)
                else:  # the folds single-partition regions still read are evaluated on their own
                    self._cp_subset[h] = np.nonzero(~mask)[0]
        self.batch_params = batch_params
        self._batch: ParamBatch | None = None
        self._batch_version = -1
        if contraction not in ("f32", "bf16x3", "bf16x6"):
            raise ValueError(f"unknown contraction {contraction!r} ('f32' = exact fp32, the product; 'bf16x3' / 'bf16x6' = labelled variants)")
        self.contraction = contraction
        self._ct = {"f32": 0, "bf16x3": 3, "bf16x6": 6}[contraction]
        self.dense_on_table = bool(dense_on_table)
        self.tiled_weights = bool(tiled_weights)
        self._assign_weight_layouts()
        if self._signed and self._tail and not self._tail16_ok():
            self._tail = []  # (only the 16-row tail walks signed values; the layers then take the complex kernels)

    def _is_real_valued(self) -> bool:
        """Every data input an Embedding layer and every parameter a plain real tensor (no parameter graph beyond the
        tensor itself): the circuit's values are real numbers, signed."""
        for l in self.layers:
            if isinstance(l, HipConstantValueLayer):
                return False
            if isinstance(l, HipInputLayer):
                if not isinstance(l, HipEmbeddingLayer):
                    return False
                ps = [l.weight]
            else:
                w = getattr(l, "weight", None)
                ps = [] if w is None else [w]
            for p in ps:
                if p.ops != ["tensor"]:
                    return False
        return not any(self.store[n].is_complex() for n in self.store.names())

    def _assign_weight_layouts(self) -> None:
        """Pick the weight layout of every K = 32 sum layer (ck_tile.h): tiled layouts only where
        the batched prologue can write them, and uniformly inside a fused launch."""
        tiled = capi.CK_W_TILED_F32
        elig = {
            i: self.batch_params and self.tiled_weights and getattr(l, "tile32_eligible", False)
            for i, l in enumerate(self.layers)
        }
        layout = {i: (tiled if ok else capi.CK_W_ROWMAJOR) for i, ok in elig.items()}
        for g in self._groups:
            members = ([g.dense_layer] if g.dense_layer is not None else []) + g.levels
            if not all(elig[j] for j in members):
                for j in members:
                    layout[j] = capi.CK_W_ROWMAJOR
        k32 = [j for j in self._tail if self.layers[j].num_output_units == 32]
        if not all(elig[j] for j in k32):
            for j in k32:
                layout[j] = capi.CK_W_ROWMAJOR
        for b in self._cp_blocks.values():  # ck_cp_lse_fwd stages row-major matrices itself
            for d in np.unique(b.slot_dense[..., 0]):
                if d >= 0:
                    layout[int(d)] = capi.CK_W_ROWMAJOR
            if b.post:
                layout[b.layer] = capi.CK_W_ROWMAJOR
        for i, l in enumerate(self.layers):
            if hasattr(l, "_w_layout"):
                l._w_layout = layout[i]

    def _group_layout(self, g: SubtreeGroup) -> int:
        members = ([g.dense_layer] if g.dense_layer is not None else []) + g.levels
        return self.layers[members[0]]._w_layout

    # -- reference surface -----------------------------------------------------------------------
    @property
    def num_variables(self) -> int:
        return self.plan.num_variables

    def __call__(self, x: torch.Tensor | None = None, *, integrate_vars=None) -> torch.Tensor:
        return self.forward(x, integrate_vars=integrate_vars)

    def __del__(self) -> None:  # pragma: no cover - interpreter teardown order
        try:
            for b in self._bindings.values():
                b.destroy()
            if self._pprog is not None:
                capi.load().ck_program_destroy(self._pprog)
        except Exception:
            pass

    # -- binding ---------------------------------------------------------------------------------
    def _layer_batch(self, l: HipLayer, B: int) -> int:
        return B

    def _bind(self, B: int) -> _Binding:
        bd = self._bindings.get(B)
        if bd is not None and bd.store_version == self.store.version:
            return bd
        if bd is not None:
            bd.destroy()
        if len(self._bindings) >= 4:  # keep a handful of batch sizes resident
            old = next(iter(self._bindings))
            self._bindings.pop(old).destroy()
        if self.cache_params:
            self._param_program()  # allocates the derived-parameter buffers the layer launches point at
        bd = _Binding()
        bd.B = B
        bd.store_version = self.store.version
        # arena layout
        bases, total = [], 0
        for i, l in enumerate(self.layers):
            bases.append(total)
            if i in self._virtual:  # fused away: never materialised
                continue
            n = l.num_folds * B * l.num_output_units
            total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        bd.arena = torch.empty(total, dtype=self._act_dtype, device=self.device)
        bd.views = [
            None
            if i in self._virtual
            else bd.arena[b : b + l.num_folds * B * l.num_output_units].view(l.num_folds, B, l.num_output_units)
            for i, (b, l) in enumerate(zip(bases, self.layers))
        ]
        # children offset tables: element offsets into the arena (replaces circuits.py:42-47)
        bd.row_off = []
        for i, (ch, l) in enumerate(zip(self._children, self.layers)):
            if ch is None or (i in self._virtual and i not in self._td_had.values()) or i in self._group_of_root:
                bd.row_off.append(None)  # (a Hadamard layer read as a list by its TensorDot layer keeps its children's offsets)
                continue
            if i in self._cp_blocks:  # slots read the dense folds' own inputs
                ch = self._cp_blocks[i].slot_child
            if i in self._regions:
                ch = self._regions[i].slot_child  # (F, H, S, 2)
            prod, fold = ch[..., 0], ch[..., 1]
            ko = np.asarray([self.layers[p].num_output_units for p in prod.reshape(-1)]).reshape(prod.shape)
            if not np.all(ko == l.num_input_units):
                raise ValueError("a layer's children do not all have its number of input units")
            off = np.asarray(bases, dtype=np.int64)[prod] + fold * (B * l.num_input_units)
            bd.row_off.append(torch.from_numpy(np.ascontiguousarray(off)).to(self.device))
        if self.plan.num_variables:  # (D, B) staging copies of the batch: fp32 and / or int32
            shape = (self.plan.num_variables, B)
            bd.xt = torch.empty(shape, dtype=torch.float32, device=self.device) if self._float_input else None
            bd.xt_i = torch.empty(shape, dtype=torch.int32, device=self.device) if self._int_input else None
        for h, folds in self._cp_subset.items():
            K = self.layers[h].num_output_units
            bd.leftover[h] = (bd.row_off[h][torch.from_numpy(folds).to(self.device)].contiguous(),
                              torch.from_numpy(bases[h] + folds * (B * K)).to(self.device))
        for d, folds in self._cp_leftover.items():  # (row offsets, output offsets) of the folds still materialised
            K = self.layers[d].num_output_units
            bd.leftover[d] = (bd.row_off[d][torch.from_numpy(folds).to(self.device)].contiguous(),
                              torch.from_numpy(bases[d] + folds * (B * K)).to(self.device))
        bd.ll = torch.empty(2, dtype=torch.float64, device=self.device)
        if self._clin is not None:
            self._clin.bind(bd)
        self._ensure_param_batch()  # (which leaf launches are persistent depends on the prologue's table jobs)
        bd.direct = self._direct_input(B)
        bd.params_at_end = self._params_at_end(B)
        bd.program = self._record(bd, with_ll=False)  # the launch list, recorded once
        # a hipGraph keeps the pointers of its capture: long launch lists read the staged copy of the batch.  The list of
        # `log_likelihood_sum` (recorded on first use, with this binding's `direct`) can be one launch longer: decide on that
        if bd.direct and self.use_graph and capi.load().ck_program_num_ops(bd.program) + 1 > self.graph_min_launches:
            capi.load().ck_program_destroy(bd.program)
            bd.direct = False
            bd.program = self._record(bd, with_ll=False)
        self._bindings[B] = bd
        return bd

    def _block_gathers(self, slot_dense: np.ndarray) -> bool:
        """Some slot of a CP block / region reads a dense layer tabulated over its categories (a staged-batch consumer)."""
        return any(int(d) in self._tdense for d in np.unique(slot_dense[..., 0]) if d >= 0)

    def _direct_input(self, B: int) -> bool:
        """Whether a forward at batch size B needs no staged copy of the discrete batch: its only readers are persistent
        leaf launches, which then read -- and validate -- the caller's int64 tensor (`ck_leaf_walk_fwd` with x_input).
        Byte offsets into the batch are 32-bit."""
        if self._clin is not None:
            # (the leaf launch of the linear-tile path CAN read and validate the int64 batch itself -- `CK_CLIN_RAW=1` when the
            #  circuit is built -- but its 8-byte row gathers cost the launch more than the staging + poison launches they save:
            #  0.248 against 0.240 ms at config 5, LAB_NOTES R6.1; the staged (D, B) copy is the default)
            return bool(self.direct_input and self._int_input and self._clin.raw_batch)
        if not (self.direct_input and self._int_input and self._groups):
            return False
        if self.plan.num_variables * B * 8 >= 2**32:
            return False
        for i, l in enumerate(self.layers):
            if i in self._virtual or (self._tail and i in self._tail):
                continue
            if i in self._group_of_root:
                if not (self._signed or self._leaf_is_persistent(self._group_of_root[i], B)):
                    return False
            elif i in self._tdense or i in self._emb_gather:
                return False
            elif i in self._cp_blocks and self._block_gathers(self._cp_blocks[i].slot_dense):
                return False
            elif i in self._regions and self._block_gathers(self._regions[i].slot_dense):
                return False
            elif isinstance(l, HipInputLayer) and not isinstance(l, HipConstantValueLayer) and not l.wants_float_input:
                return False
        return True

    def _record(self, bd: _Binding, *, with_ll: bool):
        """Record one forward for this binding; `with_ll` appends the device-side log-likelihood sum
        so that `log_likelihood_sum` replays ONE graph."""
        prog = C.c_void_p()
        capi.call("ck_program_begin", C.byref(prog))
        self._recording = True
        try:
            if not self.cache_params:
                self._enqueue_params(0, at_end=bd.params_at_end)
            self._enqueue_layers(bd, 0, with_ll=with_ll)
            if self.validate_inputs and self._int_input and not bd.direct and not self._poison_in_tail():
                for p, f in self._out_pairs:  # (complex outputs: both halves of every element)
                    v = bd.views[int(p)][int(f)]
                    capi.call("ck_poison_outputs", v.data_ptr(), v.numel() * (2 if self._complex else 1), self._bad_input.data_ptr(), 0)
            if with_ll and not self._tail_fuses_ll():
                p, f = int(self._out_pairs[0, 0]), int(self._out_pairs[0, 1])
                capi.call("ck_ll_sum", bd.views[p][f].data_ptr(), bd.B, 1, bd.ll.data_ptr(), 0)
        finally:
            self._recording = False
            capi.call("ck_program_end", prog)
        return prog

    def _raw_batch_args(self, bd: _Binding) -> tuple[int | None, int]:
        """(x_rows, x_input) of a launch that reads the caller's batch: program input cell 0 while recording, the batch of
        the last call for an eager launch (profiling)."""
        if self._recording:
            return None, 0
        if bd.x_last is None:
            raise RuntimeError("an eager launch over the raw batch needs a forward first")
        return bd.x_last.data_ptr(), -1

    def _param_program(self):
        """`cache_params`: the parameter-only launch list, recorded once per set of tensor objects
        and replayed only after a parameter VALUE changed (`TensorStore.data_version`)."""
        if self._pprog is not None and self._pprog_version == self.store.version:
            return self._pprog
        if self._pprog is not None:
            capi.load().ck_program_destroy(self._pprog)
            self._pprog = None
        prog = C.c_void_p()
        capi.call("ck_program_begin", C.byref(prog))
        try:
            self._enqueue_params(0)
        finally:
            capi.call("ck_program_end", prog)
        self._pprog, self._pprog_version, self._pprog_data_version = prog, self.store.version, None
        return prog

    def invalidate_parameters(self) -> None:
        """Tell a `cache_params` circuit that parameter values were modified in place behind the
        store's back (e.g. by an optimiser kernel writing through a raw pointer)."""
        self.store.touch()

    def _enqueue_params(self, stream: int, *, at_end: bool = False) -> None:
        """Everything that depends on the parameters only (the reference re-evaluates the parameter
        graphs on every forward, parameters/parameter.py:180-188): the batched softmax prologue, the
        remaining parameter graphs, table re-layouts, and the dense layer pushed through the table.
        `at_end`: the tail launch of the PREVIOUS forward evaluated what `_plan_tail_params` assigned to it; only the rest is
        launched here."""
        if self._clin is not None:
            return  # (plain tensors everywhere; the one derived parameter, the linear table, is the path's own first launch)
        if at_end:
            self._ensure_param_batch()
            if self._tailp["rest"] is not None:
                self._tailp["rest"].launch(stream)
        else:
            self._launch_param_batch(stream)
        for l in self.layers:
            l.prepare(stream, batched=self.batch_params)
        for g in self._groups:
            self._group_table(g, stream)

    def _scratch(self) -> torch.Tensor | None:
        """The workspace lent to the stream-K Tucker launches (`ck_set_workspace`): ticket counters (zero between
        launches) + two 16 KiB partial-tile slots per persistent workgroup.  None when no layer can use it."""
        if self._scratch_buf is None:
            tuck = [l for s, l in zip(self.plan.layers, self.layers)
                    if s.type == "tucker" and l.arity == 2 and l.num_input_units in (32, 64) and not self._complex]
            if not tuck:
                self._scratch_buf = False
            else:
                self._scratch_rows = 8  # row groups of 128 the ticket area covers (B <= 1024); grown by `_scratch_for`
                self._scratch_buf = self._new_scratch(tuck, self._scratch_rows)
                self._scratch_old: list[torch.Tensor] = []
        return None if self._scratch_buf is False else self._scratch_buf

    def _new_scratch(self, tuck, row_groups: int) -> torch.Tensor:
        tiles = max(l.num_folds * ((l.num_output_units + 31) // 32) for l in tuck) * row_groups
        nbytes = self._n_cu * 3 * 2 * (4 * 1024 + 64) * 4 + tiles * 4  # partial tiles, their (max, sum) rows, tickets
        return torch.zeros(nbytes // 4, dtype=torch.int32, device=self.device)

    def _scratch_for(self, B: int) -> torch.Tensor | None:
        """The workspace for a binding of batch size B: the ticket area grows with ceil(B / 128) (a workspace
        provisioned for B <= 1024 silently sent larger batches to the one-workgroup-per-tile launch).  Buffers that
        recorded programs of other bindings point at stay alive."""
        ws = self._scratch()
        if ws is None:
            return None
        need = (B + 127) // 128
        if need > self._scratch_rows:
            tuck = [l for s, l in zip(self.plan.layers, self.layers)
                    if s.type == "tucker" and l.arity == 2 and l.num_input_units in (32, 64) and not self._complex]
            self._scratch_old.append(self._scratch_buf)
            self._scratch_rows = need
            self._scratch_buf = self._new_scratch(tuck, need)
        return self._scratch_buf

    def _enqueue_layers(self, bd: _Binding, stream: int, *, with_ll: bool = False) -> None:
        """The layer kernels of one forward, in plan order (graph/modules.py:326-334)."""
        ws = self._scratch_for(bd.B)
        if ws is None:
            return self._enqueue_layers_(bd, stream, with_ll=with_ll)
        capi.call("ck_set_workspace", ws.data_ptr(), ws.numel() * 4)
        try:
            self._enqueue_layers_(bd, stream, with_ll=with_ll)
        finally:
            capi.call("ck_set_workspace", None, 0)

    def _flush_leftover(self, pending: list[int], bd: _Binding, stream: int) -> None:
        """The dense folds that consumers outside CP blocks still read, of SEVERAL layers, in one `ck_cp_lse_fwd` launch (a
        launch per layer of one or two folds each was five 10 us launches on BASELINE config 4's critical path): every
        fold has its own child offset, weight address and output offset, so the launch does not care which layer owns it."""
        if len(pending) == 1:
            return self._launch_cp(pending[0], bd, stream)
        key = ("leftover", tuple(pending))
        tabs = bd.cp_tabs.get(key)
        K = self.layers[pending[0]].num_output_units
        if tabs is None:
            ros, oos, ws = [], [], []
            for i in pending:
                ro, oo = bd.leftover[i]
                ros.append(ro.reshape(-1))
                oos.append(oo.reshape(-1))
                ws.append(torch.from_numpy(self.layers[i]._w.data_ptr() + self._cp_leftover[i].astype(np.int64) * (K * K * 4)).to(self.device))
            tabs = bd.cp_tabs[key] = (torch.cat(ros).contiguous(), torch.cat(ws).contiguous(), torch.cat(oos).contiguous())
        ro, tab, oo = tabs
        capi.call("ck_cp_lse_fwd", bd.arena.data_ptr(), ro.data_ptr(), tab.data_ptr(), None, oo.data_ptr(),
                  bd.arena.data_ptr(), None, None, None, 0, int(ro.numel()), 1, 1, bd.B, K, stream)

    def _enqueue_layers_(self, bd: _Binding, stream: int, *, with_ll: bool = False) -> None:
        B = bd.B
        if self._clin is not None:
            self._clin.enqueue(bd, stream)
            return
        pending: list[int] = []  # leftover dense folds (`_cp_leftover`) not launched yet: they wait for their first reader
        for i, (l, view, ro) in enumerate(zip(self.layers, bd.views, bd.row_off)):
            if pending and self._children[i] is not None and set(int(p) for p in np.unique(self._children[i][..., 0])) & set(pending):
                self._flush_leftover(pending, bd, stream)
                pending = []
            if self._tail and i in self._tail:
                if i == self._tail[0]:
                    if pending:
                        self._flush_leftover(pending, bd, stream)
                        pending = []
                    self._launch_tail(bd, stream, with_ll=with_ll)
                continue
            if i in self._virtual or i in self._td_first:
                continue
            if i in self._td_had or i in self._td_pair:
                self._launch_tensordot(i, bd, stream)
            elif i in self._group_of_root:
                self._launch_group(self._group_of_root[i], bd, view, stream)
            elif i in self._tdense:
                self._launch_table_dense(i, bd, stream)
            elif i in self._emb_gather:
                self._launch_emb_gather(i, bd, stream)
            elif i in self._cp_leftover and i not in self._cp_blocks and not self._complex and (
                    not pending or self.layers[pending[0]].num_output_units == l.num_output_units):
                pending.append(i)  # (launched together with the other leftovers, before the first layer that reads one)
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
        if pending:
            self._flush_leftover(pending, bd, stream)

    # -- evaluation ------------------------------------------------------------------------------
    def _prepare_input(self, x: torch.Tensor) -> tuple[torch.Tensor | None, torch.Tensor | None]:
        """Device / dtype conversions of the (B, D) batch (torch ops on the CURRENT stream, so they
        must be issued before the evaluation stream waits on it).  Returns the float32 and the int64
        view of the batch, whichever the input layers need (both for mixed continuous / discrete
        inputs; a float batch is truncated like ``x.long()``, input.py:400-401, and a NaN -- the
        marginalisation sentinel of the continuous layers -- becomes the discrete sentinel -1)."""
        if x.device != self.device:
            x = x.to(self.device)
        if x.shape[1] != self.plan.num_variables:
            x = x[:, : self.plan.num_variables]
        xf = xi = None
        if self._float_input:
            xf = x.to(torch.float32).contiguous()
        if self._int_input:
            if x.is_floating_point():
                x = torch.where(torch.isnan(x), torch.full((), -1.0, device=x.device, dtype=x.dtype), x)
            xi = x.to(torch.int64).contiguous()
        return xf, xi

    def _stage_input(self, bd: _Binding, xf, xi, stream: int) -> None:
        """(B, D) batch -> (D, B) staging copies (replaces circuits.py:66)."""
        if xf is not None:
            capi.call("ck_transpose_f32", xf.data_ptr(), bd.xt.data_ptr(), bd.B, self.plan.num_variables, stream)
        if xi is not None:
            if self.validate_inputs:
                capi.call("ck_stage_categories", xi.data_ptr(), bd.xt_i.data_ptr(), bd.B, self.plan.num_variables,
                          self._num_states_dev().data_ptr(), self._bad_input.data_ptr(), 1 if self._preclamp() else 0, None, stream)
            else:
                capi.call("ck_transpose_i64_to_i32", xi.data_ptr(), bd.xt_i.data_ptr(), bd.B, self.plan.num_variables, stream)

    def _preclamp(self) -> bool:
        """Whether the staging kernel writes table row numbers (the range mapping every consumer would apply) instead of
        raw values: possible when every discrete layer reading a variable indexes it with the same number of states."""
        self._num_states_dev()
        return self.validate_inputs and self._states_consistent

    def _num_states_dev(self) -> torch.Tensor:
        """(D,) int32: number of states the discrete input layers index variable d with (the smallest, if several layers
        read it; 0 = no discrete layer reads it)."""
        if self._num_states is None:
            ns = np.zeros(max(1, self.plan.num_variables), dtype=np.int64)
            self._states_consistent = True
            for l in self.layers:
                if not isinstance(l, HipInputLayer) or isinstance(l, HipConstantValueLayer) or l.wants_float_input:
                    continue
                n = getattr(l, "num_categories", None) or getattr(l, "num_states", None)
                if n is None and hasattr(l, "total_count"):
                    n = int(l.total_count) + 1
                if n is None:
                    continue
                for v in np.unique(l.scope_idx):
                    if ns[v] not in (0, n):
                        self._states_consistent = False
                    ns[v] = n if ns[v] == 0 else min(ns[v], n)
            self._num_states = torch.from_numpy(ns.astype(np.int32)).to(self.device)
        return self._num_states

    def check_inputs(self) -> None:
        """Raise ``IndexError`` if a batch evaluated since the last check held a category outside its layer's range
        (what ``TorchCategoricalLayer`` / ``TorchEmbeddingLayer`` raise from their advanced indexing, input.py:258-266,
        399-412).  The forward itself never waits for the device: an invalid batch makes the circuit's outputs NaN (the
        flag is sticky, like a device-side assert) and this call -- which synchronises -- says why and clears it."""
        if not self.validate_inputs:
            return
        if int(self._bad_input.item()) != 0:
            self._bad_input.zero_()
            raise IndexError("a batch held a category index out of range for its input layer "
                             "(outputs are NaN from that batch on until this check)")

    def _apply_integration_mask(self, x: torch.Tensor, integrate_vars) -> torch.Tensor:
        """Marginalisation (IntegrateQuery, cirkit/backend/torch/queries.py:19-184): a boolean mask
        ``(B, D)`` / ``(1, D)`` / ``(D,)`` or an iterable of variable ids.  Masked entries are replaced by
        the sentinel the input kernels understand (negative category / NaN), which makes them emit
        the layer's integral instead of a likelihood."""
        D = self.plan.num_variables
        if not isinstance(integrate_vars, torch.Tensor):
            ids = sorted(int(v) for v in integrate_vars)
            if ids and (ids[0] < 0 or ids[-1] >= D):
                raise ValueError("The variables to marginalize must be a subset of the circuit scope")
            mask = torch.zeros((1, D), dtype=torch.bool)
            mask[0, ids] = True
        else:
            mask = integrate_vars
            if mask.dtype != torch.bool:
                raise ValueError(f"Expected dtype of tensor to be torch.bool, got {mask.dtype}")
            if mask.dim() == 1:
                mask = mask.unsqueeze(0)
            if mask.shape[1] != D:
                raise ValueError(f"Circuit scope has {D} variables but integrate_vars was defined over "
                                 f"{mask.shape[1]} != {D} variables")
        if mask.shape[0] not in (1, x.shape[0]):
            raise ValueError("The number of scopes to integrate over must either match the batch size of x, or be 1")
        for l in self.layers:
            if isinstance(l, HipInputLayer) and not l.can_integrate:
                raise NotImplementedError(f"marginalisation through {type(l).__name__}")
        mask = mask.to(x.device)
        if self._float_input:  # (the discrete layers of a mixed circuit see NaN as -1, _prepare_input)
            return torch.where(mask, torch.full((), float("nan"), device=x.device, dtype=torch.float32), x.to(torch.float32))
        return torch.where(mask, torch.full((), -1, device=x.device, dtype=torch.int64), x.to(torch.int64))

    def _run(self, x: torch.Tensor | None, *, with_ll: bool = False, ll_out: torch.Tensor | None = None) -> _Binding:
        if self.plan.num_variables:
            if x is None:
                raise ValueError(f"Expected some input 'x', as the circuit has {self.plan.num_variables} variables")
            if x.dim() != 2:
                raise ValueError(
                    "The input to the circuit should have shape (B, D), where B is the batch size and D "
                    "is the number of variables the circuit is defined on"
                )
            if x.shape[1] < self.plan.num_variables:
                raise ValueError(f"expected at least {self.plan.num_variables} variables, found {x.shape[1]}")
            B = int(x.shape[0])
        else:
            B = 1 if x is None else int(x.shape[0])
        if B <= 0:
            raise ValueError("empty batch")
        bd = self._bind(B)
        with torch.cuda.device(self.device):
            xf, xi = self._prepare_input(x) if self.plan.num_variables else (None, None)
            cur = torch.cuda.current_stream(self.device)
            run = cur
            if with_ll and bd.program_ll is None:
                bd.program_ll = self._record(bd, with_ll=True)
            prog = bd.program_ll if with_ll else bd.program
            lib = capi.load()
            as_graph = bool(self.use_graph) and lib.ck_program_num_ops(prog) > self.graph_min_launches
            state = self.store.state() if (self.cache_params or bd.params_at_end) else None
            refresh = self.cache_params and self._pprog_data_version != state
            pprog = self._param_program() if refresh else None
            p_graph = refresh and bool(self.use_graph) and lib.ck_program_num_ops(pprog) > self.graph_min_launches
            if (as_graph or p_graph) and cur.cuda_stream == 0:  # a capture cannot run on the legacy default stream
                if self._side is None:
                    self._side = torch.cuda.Stream(self.device)
                run = self._side
                run.wait_stream(cur)
            stream = run.cuda_stream
            if self.plan.num_variables:
                self._stage_input(bd, xf, None if bd.direct else xi, stream)
            if bd.direct:  # the recorded leaf / tail launches read the pointer from input cell 0 at replay
                if as_graph:
                    raise RuntimeError("a binding that reads the raw batch cannot be replayed as a hipGraph")
                bd.x_last = xi
                capi.call("ck_program_set_input", prog, 0, xi.data_ptr())
            if with_ll and bd.ll_cell:  # (an eager replay reads the cell; NULL = the binding's own pair)
                direct_ll = ll_out is not None and not as_graph
                capi.call("ck_program_set_input", prog, bd.ll_cell, ll_out.data_ptr() if direct_ll else None)
                if direct_ll:
                    ll_out = None
            if refresh:
                self._pprog_data_version = state
                capi.call("ck_program_launch", pprog, 1 if p_graph else 0, stream)
            if bd.params_at_end:
                # the launch that ends this forward re-evaluates the parameters for the next one; a store that has changed
                # since the derived parameters in memory were evaluated gets them evaluated now, on their own
                if self._params_valid_version != state:
                    self._launch_param_batch(stream)
                self._params_valid_version = state
            capi.call("ck_program_launch", prog, 1 if as_graph else 0, stream)
            if run is not cur:
                cur.wait_stream(run)
            if ll_out is not None:  # no launch of this binding takes the destination: one 16-byte copy
                ll_out.copy_(bd.ll)
        return bd

    def replays_as_graph(self, B: int, *, with_ll: bool = False) -> bool:
        """Whether a forward of batch size B is replayed as a hipGraph (long launch lists) or eagerly by the native
        executor (short ones, see `use_graph`)."""
        bd = self._bind(B)
        if with_ll and bd.program_ll is None:
            bd.program_ll = self._record(bd, with_ll=True)
        prog = bd.program_ll if with_ll else bd.program
        return bool(self.use_graph) and capi.load().ck_program_num_ops(prog) > self.graph_min_launches

    def forward(self, x: torch.Tensor | None = None, *, integrate_vars=None) -> torch.Tensor:
        """Returns ``(B, O, K)`` like ``TorchCircuit.forward`` (``(O, K)`` for an empty-scope circuit).
        With ``integrate_vars`` the listed / masked variables are marginalised out
        (``IntegrateQuery.__call__``).  The result aliases the circuit's arena: it is overwritten by
        the next call with the same batch size (clone it to keep it)."""
        if integrate_vars is not None:
            if x is None:
                raise ValueError("integrate_vars needs an input batch")
            x = self._apply_integration_mask(x.to(self.device), integrate_vars)
        bd = self._run(x)
        pairs = self._out_pairs
        if len(pairs) == 1:
            p, f = int(pairs[0, 0]), int(pairs[0, 1])
            y = bd.views[p][f : f + 1]  # (1, B, K)
        else:
            y = torch.stack([bd.views[int(p)][int(f)] for p, f in pairs], dim=0)  # (O, B, K)
        y = y.transpose(0, 1)
        if self._pad_info is not None and y.shape[-1] != self._pad_info.out_units:
            y = y[..., : self._pad_info.out_units]
        if self.plan.num_variables == 0:
            y = y.squeeze(0)
        return y

    def layer_outputs(self, x: torch.Tensor | None = None) -> list[torch.Tensor]:
        """All ``(F, B, Ko)`` layer outputs of one forward (views of the arena; None for layers that
        cross-layer fusion never materialises, or materialises only in part) -- for parity tests."""
        views = list(self._run(x).views)
        for d in list(self._cp_leftover) + list(self._cp_subset):
            views[d] = None
        if self._pad_info is not None:
            views = [v if v is None else v[..., : s.num_output_units] for v, s in zip(views, self.user_plan.layers)]
        return views

    def log_likelihood_sum(self, x: torch.Tensor, out: torch.Tensor | None = None, *, reduce: bool = False) -> torch.Tensor:
        """Device tensor ``[sum_b log p(x_b), B]`` in fp64 -- the two numbers the data-parallel
        all-reduce exchanges (SURVEY.md section 8 e).  Requires a single scalar output.

        `reduce`: SUM the pair over the data-parallel ranks, in place, right behind the forward on the same stream
        (`cirkit_amd.distributed.all_reduce_sum`: RCCL through the C ABI when a `HipComm` is set, torch.distributed otherwise).

        `out`: a contiguous fp64 tensor of two elements on this device (e.g. one row of a (steps, 2) buffer that a single
        collective will carry) that receives the pair and is returned; the launch that ends the forward writes it there
        itself where it can (ck_tail_params_fwd's ll_cell), a 16-byte copy does otherwise.  Without `out` the result is the
        binding's own buffer, overwritten by the next call at this batch size."""
        pairs = self._out_pairs
        if len(pairs) != 1 or self._complex:
            raise ValueError("log_likelihood_sum needs a real circuit with one output")
        if self.layers[int(pairs[0, 0])].num_output_units != 1:
            raise ValueError("log_likelihood_sum needs a scalar output unit")
        if out is not None:
            index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            if out.dtype != torch.float64 or out.numel() != 2 or not out.is_contiguous() or not out.is_cuda or out.device.index != index:
                raise ValueError(f"out must be a contiguous float64 tensor of 2 elements on {self.device}")
            self._run(x, with_ll=True, ll_out=out)
            if reduce:
                from .distributed import all_reduce_sum

                all_reduce_sum(out)
            return out
        ll = self._run(x, with_ll=True).ll
        if reduce:
            from .distributed import all_reduce_sum

            all_reduce_sum(ll)
        return ll

    # -- accounting ------------------------------------------------------------------------------
    def arena_bytes(self, B: int) -> int:
        esz = 8 if self._complex else 4
        return sum(l.num_folds * B * l.num_output_units for l in self.layers) * esz

    def num_launches(self, B: int) -> int:
        return int(capi.load().ck_program_num_ops(self._bind(B).program))

    def num_launches_ll(self, B: int) -> int:
        """Launches of one `log_likelihood_sum` step: the recorded program plus the staging of the batch in front of it
        (none when the leaf launches read the caller's batch, `direct_input`)."""
        bd = self._bind(B)
        if bd.program_ll is None:
            bd.program_ll = self._record(bd, with_ll=True)
        staging = 0 if bd.direct else int(self._float_input) + int(self._int_input)
        return int(capi.load().ck_program_num_ops(bd.program_ll)) + (staging if self.plan.num_variables else 0)

    def reads_batch_directly(self, B: int) -> bool:
        return self._bind(B).direct


class HipCircuitStreams:
    """Several forwards in flight: `n` circuits over the SAME parameter store (own activation arenas
    and derived parameters), calls dealt round-robin to `n` HIP streams.

    A forward of a small circuit is a short chain of kernels of which only one fills the GPU (at
    BASELINE config 2: 98 of 171 us); the parameter prologue and the few-fold tail are latency-bound.
    With two forwards in flight those fill each other's bubbles: 31 M instead of 24 M evaluations/s at
    config 2 (`bench.py`, `variants["streams=2"]`).  Per-call latency is unchanged, and every result is
    the complete forward of its own batch.

    `forward` / `log_likelihood_sum` return the result together with the stream it is being computed on;
    consume it on that stream or after `synchronize()`."""

    def __init__(self, plan: Plan, tensors, *, n: int = 2, device: str | torch.device = "cuda:0",
                 wait_for_input: bool = True, **kwargs) -> None:
        """`wait_for_input`: make the chosen stream wait for the work already queued on the caller's current
        stream (where `x` is presumably produced).  An event record + wait per call costs a few microseconds;
        pass False when the inputs are known to be ready (e.g. resident batches)."""
        if n < 1:
            raise ValueError("n must be at least 1")
        self.wait_for_input = bool(wait_for_input)
        first = HipCircuit(plan, tensors, device=device, **kwargs)
        self.circuits = [first] + [HipCircuit(plan, first.store, device=device, **kwargs) for _ in range(n - 1)]
        self.streams = [torch.cuda.Stream(first.device) for _ in range(n)]
        self.device = first.device
        self.store = first.store
        self._next = 0

    def _take(self):
        i = self._next
        self._next = (i + 1) % len(self.circuits)
        return self.circuits[i], self.streams[i]

    def forward(self, x: torch.Tensor | None = None, *, integrate_vars=None):
        c, st = self._take()
        if self.wait_for_input:
            st.wait_stream(torch.cuda.current_stream(self.device))  # x may still be in production
        with torch.cuda.stream(st):
            return c.forward(x, integrate_vars=integrate_vars), st

    __call__ = forward

    def log_likelihood_sum(self, x: torch.Tensor):
        c, st = self._take()
        if self.wait_for_input:
            st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            return c.log_likelihood_sum(x), st

    def synchronize(self) -> None:
        for st in self.streams:
            st.synchronize()
