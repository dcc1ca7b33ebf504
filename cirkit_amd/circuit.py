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
from typing import Mapping

import numpy as np
import torch

from . import _capi as capi
from . import padding
from .fusion import (CPBlock, RegionBlock, SubtreeGroup, find_cp_blocks, find_input_products, find_region_blocks,
                     find_subtree_groups, find_table_dense, find_tail, leaf_segments)
from .layers import HipConstantValueLayer, HipEmbeddingLayer, HipInputLayer, HipLayer, HipTensorDotLayer, layer_from_spec
from .parameters import ParamBatch, TensorStore
from .plan import Plan, resolve_fold_index

# ck_tail16_fold of include/cirkit_hip.h
_TAIL16_FOLD = np.dtype([("w", "<u8"), ("out", "<u8"), ("child", "<u8", (4,)), ("child_src", "<i4", (4,)), ("H", "<i4"),
                         ("Ko", "<i4"), ("skip_store", "<i4"), ("slot", "<i4")])
assert _TAIL16_FOLD.itemsize == 80

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


class HipCircuit:
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
        self._groups: list[SubtreeGroup] = (
            find_subtree_groups(plan, self.layers, self._children, self._out_pairs, depth, signed=self._signed)
            if fuse is not False else []
        )
        if self._signed and (not self._groups or any(g.depth < 1 or self.layers[g.input_layer].num_states >= 65535 for g in self._groups)):
            self._signed, self._groups = False, []
        self._group_of_root = {g.root: g for g in self._groups}
        self._virtual = {i for g in self._groups for i in g.virtual}
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
                          self._num_states_dev().data_ptr(), self._bad_input.data_ptr(), 1 if self._preclamp() else 0, stream)
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
            return f"leaf_persistent_kernel<{self._group_of_root[i].depth}, 8, true, {raw}, false, false, 0> (signed: real-valued complex circuit)"
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
