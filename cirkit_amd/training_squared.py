"""Plan-level training of squared circuits p(x) = |c(x)|^2 / Z with real parameters, under complex-lse-sum or lse-sum.

The reference trains such a model with autograd through its torch layers (``loss = -mean(2 Re c(x) - Re Z)``: c compiled under
complex-lse-sum -- Embedding + CP-T / sum layers, layers/input.py:258-266, optimized.py:171-178, semiring.py:441-476 -- and
Z = integrate(multiply(c, conj(c))) made of ConstantValue, Hadamard and TensorDot layers whose parameters are pointer / conj /
einsum / flatten graphs over the tensors of c, symbolic/operators.py:39-322; a REAL circuit squares the same way under lse-sum,
``loss = -mean(2 c(x) - Z)``, Categorical inputs whose products integrate to logs of Gram matrices, operators.py:51-63,
106-139).  `cirkit_amd.training.HipTrainer` covers single real circuits; this module is its counterpart for that model class:

* forward: two layer-wise `HipCircuit`s (c on the batch, Z on no input) sharing ONE parameter store;
* backward: a reverse launch list per circuit over the gradient arena (complex64 or fp32) -- `ck_sum_lse_bwd_c` /
  `ck_sum_lse_bwd` for sum / CP-T / Tucker layers and, on a permuted copy of their input, TensorDot layers; `ck_hadamard_bwd`
  (on (re, im) pairs under the complex semiring); `ck_categorical_bwd` + d log w / dw for Embedding layers, +
  `ck_param_log_table_bwd` for Categorical layers; the batch sum (+ d log v / dv) for ConstantValue layers -- and
  `HipParameter.backward` through the parameter graphs (pointer gathers, conj of real values, softmax, einsum, log, flatten:
  cirkit_amd/parameters.py);
* one flat parameter / gradient / moment buffer: one optimizer launch, one all-reduce.

torch appears as storage only: after the forwards of c and Z (two recorded programs each) the backward lists, the
log-likelihood pair and -- on one rank -- the optimizer with its device clock (`ck_opt_state`) are three recorded `ck_program`s per
batch size (c, Z beside it on a second stream, the end), replayed by the native executor (`use_graph=True`: as hipGraphs -- measured
slower: 1.30 against 1.24 ms per 4096 rows); no tensor-library kernel and no allocation in a step.  Restrictions (checked, `NotImplementedError`): real parameter tensors, every (layer, fold)
read by exactly one consumer (trees: what `squared_partition_plan` and the region-graph templates give), a scalar output."""
from __future__ import annotations

import ctypes as C
import os
from typing import Mapping

import numpy as np
import torch

from .distributed import all_reduce_sum as _all_reduce_sum, default_comm as _default_comm, world_size as _world_size
from . import _capi as capi
from .circuit import HipCircuit
from .layers import (HipCategoricalLayer, HipConstantValueLayer, HipCPTLayer, HipEmbeddingLayer, HipGaussianLayer, HipHadamardLayer,
                     HipSumLayer, HipTensorDotLayer, HipTuckerLayer)
from .parameters import TensorStore
from .plan import Plan


class _PlanBackward:
    """The reverse launch list of ONE layer-wise circuit (complex-lse-sum or lse-sum) over the activations of its last forward.
    Every buffer a launch touches is allocated when a batch size is bound, `run` issues calls of the C ABI only: the list can be
    recorded into a `ck_program` (HipSquaredTrainer does) and holds no tensor-library kernel."""

    def __init__(self, circuit: HipCircuit, grads: Mapping[str, torch.Tensor]) -> None:
        self.c, self.grads = circuit, grads
        self.cplx = circuit.plan.semiring == "complex-lse-sum"
        self._bound: dict[int, dict] = {}
        c = circuit
        if len(c._out_pairs) != 1:
            raise NotImplementedError("training needs a single circuit output")
        po = int(c._out_pairs[0, 0])
        if c.layers[po].num_output_units != 1:
            raise NotImplementedError("training needs a scalar output unit")
        seen: set[tuple[int, int]] = set()
        for j, ch in enumerate(c._children):
            if ch is None:
                continue
            for p, f in ch.reshape(-1, 2):
                if (int(p), int(f)) in seen:
                    raise NotImplementedError("squared-circuit training: a fold read by several consumers (gradients are stored, not added)")
                seen.add((int(p), int(f)))
        for spec, l in zip(c.plan.layers, c.layers):
            if isinstance(l, (HipSumLayer, HipCPTLayer, HipTuckerLayer)):
                if getattr(l, "_mixing", False):
                    raise NotImplementedError("squared-circuit training: mixing layers")
            elif isinstance(l, HipCategoricalLayer):
                if self.cplx or l.probs is None or l.probs.softmax_source() is None:
                    raise NotImplementedError("squared-circuit training: Categorical layers need lse-sum and probs = softmax(tensor)")
            elif isinstance(l, HipGaussianLayer):
                if self.cplx or l.log_partition is not None:
                    raise NotImplementedError("squared-circuit training: Gaussian layers need lse-sum and no log-partition parameter")
            elif not isinstance(l, (HipEmbeddingLayer, HipConstantValueLayer, HipHadamardLayer, HipTensorDotLayer)):
                raise NotImplementedError(f"squared-circuit training: layer type {spec.type!r}")
        # What the TensorDot launches take over (cirkit_amd/fusion.py: Hadamard layers read as lists, the W / conj W pair of
        # TensorDot layers in one launch)
        from .fusion import tensordot_lists

        self.had_of, self.pair_of = tensordot_lists(c.layers, c._children, {po})
        self._skip = set(self.had_of.values()) | set(self.pair_of.values())

    def _bind(self, B: int) -> dict:
        bd = self.c._bind(B)
        st = self._bound.get(B)
        if st is not None and st["arena_ptr"] == bd.arena.data_ptr():
            return st
        dev = bd.arena.device
        cplx = self.cplx
        e = 2 if cplx else 1  # floats per value
        garena = torch.zeros_like(bd.arena)  # complex64 / fp32, the mirror of the activation arena
        gviews = []
        for i, l in enumerate(self.c.layers):
            off = (bd.views[i].data_ptr() - bd.arena.data_ptr()) // bd.arena.element_size()
            gviews.append(garena[off : off + l.num_folds * B * l.num_output_units].view(l.num_folds, B, l.num_output_units))

        def f32(*shape):
            return torch.zeros(shape, dtype=torch.float32, device=dev)

        # per layer: the scratch of its backward (gradients of its evaluated parameters, permuted copies, gather tables), the
        # child offsets on the host (read once: a `.tolist()` per step would be a device synchronisation per layer) and in float units
        scratch: dict[int, dict] = {}
        for i, l in enumerate(self.c.layers):
            F, K = l.num_folds, l.num_output_units
            sc: dict = {}
            if isinstance(l, HipTensorDotLayer):
                pass  # (d w: a slice of the pool, or the tensor's gradient itself, `_weight_pool`)
            elif isinstance(l, (HipSumLayer, HipCPTLayer, HipTuckerLayer)):
                pass  # (d w: a slice of the pool, `_weight_pool`)
            elif isinstance(l, HipHadamardLayer):
                sc["ro"] = (bd.row_off[i] * e).contiguous()
            elif isinstance(l, HipCategoricalLayer):
                sc["dtable"] = f32(F, l.num_categories + 1, K)
            elif isinstance(l, HipGaussianLayer):
                sc["dm"], sc["dsd"] = f32(F, K), f32(F, K)
            elif isinstance(l, HipEmbeddingLayer):
                sc["dw"] = f32(F, K, l.num_states)
                words = (l.num_states + 1) * (K + 1) + 2 * l.num_states + 3 + 4096
                sc["one_launch"] = K % 32 == 0 and words * 4 <= 160 * 1024  # (ck_embedding_bwd's conditions)
                if not sc["one_launch"]:
                    sc["dtable"] = f32(F, l.num_states + 1, K)
                    if cplx:
                        sc["gr"] = f32(F, B, K)
            elif isinstance(l, HipConstantValueLayer):
                if cplx or B > 1:
                    sc["gsum"] = f32(F, K)
                if B > 1:
                    sc["ones"] = torch.ones((F, 1, B), dtype=torch.float32, device=dev)
                    if cplx:
                        sc["gr"] = f32(F, B, K)
                if not l.log_space:
                    sc["dv"] = f32(F, K)
            scratch[i] = sc
        st = {"arena_ptr": bd.arena.data_ptr(), "garena": garena, "gviews": gviews, "scratch": scratch, "pool": None}
        while len(self._bound) >= 4:
            self._bound.pop(next(iter(self._bound)))
        self._bound[B] = st
        return st

    def _td_args(self, i: int, bd):
        """(row offsets, list length) of TensorDot layer i's input: its own block, or the children of the Hadamard layer it absorbs."""
        h = self.had_of.get(i)
        return (bd.row_off[i].data_ptr(), 1) if h is None else (bd.row_off[h].data_ptr(), self.c.layers[h].arity)

    def forward(self, B: int, stream: int) -> None:
        """The layer launches of a circuit WITHOUT input variables (the partition function: `HipCircuit._enqueue_layers` with the
        Hadamard layers read as lists and the TensorDot pairs in one launch each); the parameters must have been evaluated."""
        c = self.c
        if c.plan.num_variables:
            raise NotImplementedError("_PlanBackward.forward: circuits with input variables go through HipCircuit")
        bd = c._bind(B)
        aa, cv = bd.arena.data_ptr(), 1 if self.cplx else 0
        for i, l in enumerate(c.layers):
            if i in self._skip:
                continue
            if isinstance(l, HipTensorDotLayer):
                if l._w.is_complex():
                    raise NotImplementedError("squared-circuit training: complex-valued weights")
                a = self.pair_of.get(i)
                if a is None:
                    ro, H = self._td_args(i, bd)
                    capi.call("ck_tensordot_lse_fwd_h", aa, ro, H, l._w.data_ptr(), bd.views[i].data_ptr(), l.num_folds, B,
                              l._num_contract_units, l._num_batch_units, l.num_output_units // l._num_batch_units, cv, stream)
                else:
                    la = c.layers[a]
                    ro, H = self._td_args(a, bd)
                    capi.call("ck_tensordot2_lse_fwd", aa, ro, H, la._w.data_ptr(), bd.views[a].data_ptr(), l._w.data_ptr(), bd.views[i].data_ptr(),
                              l.num_folds, B, la._num_contract_units, la._num_batch_units, la.num_output_units // la._num_batch_units,
                              l.num_output_units // l._num_batch_units, cv, stream)
            elif isinstance(l, HipConstantValueLayer):
                l.launch_const(bd.views[i], B, stream)
            else:
                l.launch(bd.arena, bd.row_off[i], bd.views[i], B, stream)

    def _weight_pool(self, st: dict) -> torch.Tensor:
        """The gradients of the evaluated weights of every sum / TensorDot layer as slices of ONE buffer (the kernels add into
        them with atomics: one fill per step zeroes them all); sized at the first run, when the forward has evaluated the weights."""
        if st["pool"] is None:
            shapes, off = {}, 0
            for i, l in enumerate(self.c.layers):
                if isinstance(l, HipTensorDotLayer):
                    shape = (l.num_folds, l.num_output_units // l._num_batch_units, l._num_contract_units)
                elif isinstance(l, (HipSumLayer, HipCPTLayer, HipTuckerLayer)):
                    shape = tuple(l._w.shape)
                else:
                    continue
                # a weight that IS a stored tensor (a pointer, through conjugates): the kernel adds into that tensor's gradient
                name = l.weight.passthrough_tensor()
                if name is not None and self.grads[name].numel() == int(np.prod(shape)) and self.grads[name].is_contiguous():
                    st["scratch"][i]["dw"] = self.grads[name].view(shape)
                    st["scratch"][i]["direct"] = True
                    continue
                shapes[i] = (off, shape)
                off += (int(np.prod(shape)) + 3) // 4 * 4
            pool = torch.zeros(max(off, 4), dtype=torch.float32, device=st["garena"].device)
            for i, (o, shape) in shapes.items():
                st["scratch"][i]["dw"] = pool[o : o + int(np.prod(shape))].view(shape)
                st["scratch"][i]["direct"] = False
            st["pool"] = pool
        return st["pool"]

    def zero_pool(self, B: int, stream: int) -> None:
        """Zero the weight-gradient pool of this binding (what `run(..., pool_zeroed=True)` then adds into)."""
        pool = self._weight_pool(self._bind(B))
        capi.call("ck_fill_f32", pool.data_ptr(), pool.numel(), 0.0, stream)

    def run(self, B: int, seed_real: float, stream: int, pool_zeroed: bool = False) -> None:
        """Gradients of ``seed_real * sum_b Re out_b`` w.r.t. the parameter tensors, ADDED into `grads`."""
        c = self.c
        bd = c._bind(B)
        st = self._bind(B)
        garena, gviews = st["garena"], st["gviews"]
        cplx = self.cplx
        e = 2 if cplx else 1
        esz = 4 * e
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        if st.get("seed_key") != float(seed_real):  # (the output layer's gradient is read, never written: once per binding and seed)
            capi.call("ck_fill_f32", gviews[po].data_ptr(), gviews[po].numel() * e, 0.0, stream)
            capi.call("ck_fill_strided_f32", gviews[po][fo].data_ptr(), B, e, float(seed_real), stream)  # (Re = seed, Im = 0)
            st["seed_key"] = float(seed_real)
        ga, aa = garena.data_ptr(), bd.arena.data_ptr()
        pool = self._weight_pool(st)
        if not pool_zeroed:
            capi.call("ck_fill_f32", pool.data_ptr(), pool.numel(), 0.0, stream)

        def sum_bwd(arena_ptr, garena_ptr, row_off, w, out_ptr, g_ptr, dw, F, H, rows, Ki, Ko, mode):
            if w.is_complex():
                raise NotImplementedError("squared-circuit training: complex-valued weights")
            if cplx:
                capi.call("ck_sum_lse_bwd_c", arena_ptr, garena_ptr, row_off.data_ptr(), w.data_ptr(), out_ptr, g_ptr, dw.data_ptr(),
                          F, H, rows, Ki, Ko, mode, 0, stream)
            else:
                capi.call("ck_sum_lse_bwd", arena_ptr, garena_ptr, row_off.data_ptr(), None, w.data_ptr(), out_ptr, g_ptr, dw.data_ptr(),
                          F, H, rows, Ki, Ko, mode, 0, stream)

        def real_part(g, sc):  # (F, B, K) fp32: the real parts of a gradient block
            if not cplx:
                return g
            capi.call("ck_copy_strided_f32", g.data_ptr(), sc["gr"].data_ptr(), g.numel(), 2, 1, stream)
            return sc["gr"]

        for i in range(len(c.layers) - 1, -1, -1):
            l = c.layers[i]
            F, K = l.num_folds, l.num_output_units
            g = gviews[i]
            sc = st["scratch"][i]
            if i in self._skip:  # (its reader's launch did its part: an absorbed Hadamard layer, the first half of a pair)
                continue
            if isinstance(l, HipTensorDotLayer):  # (optimized.py:289-296) on its own layout: x (B, Kj, Kq) -> out (B, Kq, Kk)
                if l._w.is_complex():
                    raise NotImplementedError("squared-circuit training: complex-valued weights")
                cv = 1 if cplx else 0
                a = self.pair_of.get(i)
                if a is None:
                    ro, H = self._td_args(i, bd)
                    capi.call("ck_tensordot_lse_bwd", aa, ga, ro, H, l._w.data_ptr(), bd.views[i].data_ptr(), g.data_ptr(), sc["dw"].data_ptr(),
                              F, B, l._num_contract_units, l._num_batch_units, K // l._num_batch_units, cv, stream)
                else:
                    la, sa = c.layers[a], st["scratch"][a]
                    if la._w.is_complex():
                        raise NotImplementedError("squared-circuit training: complex-valued weights")
                    ro, H = self._td_args(a, bd)
                    capi.call("ck_tensordot2_lse_bwd", aa, ga, ro, H, la._w.data_ptr(), bd.views[a].data_ptr(), gviews[a].data_ptr(),
                              l._w.data_ptr(), bd.views[i].data_ptr(), g.data_ptr(), sa["dw"].data_ptr(), sc["dw"].data_ptr(), F, B,
                              la._num_contract_units, la._num_batch_units, la.num_output_units // la._num_batch_units, K // l._num_batch_units,
                              cv, stream)
                    if not sa["direct"]:
                        la.weight.backward(sa["dw"], self.grads, stream)
                if not sc["direct"]:
                    l.weight.backward(sc["dw"], self.grads, stream)
            elif isinstance(l, (HipSumLayer, HipCPTLayer, HipTuckerLayer)):
                w = l._w
                sum_bwd(aa, ga, bd.row_off[i], w, bd.views[i].data_ptr(), g.data_ptr(), sc["dw"], F, l.arity, B, l.num_input_units, K, l._mode)
                if not sc["direct"]:
                    l.weight.backward(sc["dw"], self.grads, stream)
            elif isinstance(l, HipHadamardLayer):  # log space: the sum of the children -- (re, im) pairs as 2 K floats
                capi.call("ck_hadamard_bwd", ga, sc["ro"].data_ptr(), g.data_ptr(), F, l.arity, B, e * K, 0, stream)
            elif isinstance(l, HipCategoricalLayer):  # (lse-sum) the scatter-add into the log-table, then log softmax backward
                Cn = l.num_categories
                dtable = sc["dtable"]
                capi.call("ck_categorical_bwd", g.data_ptr(), None, bd.xt_i.data_ptr(), l._scope(g.device).data_ptr(), dtable.data_ptr(),
                          F, B, K, Cn, 0, None, stream)
                name = l.probs.graph.nodes[0].config["tensor"]
                capi.call("ck_param_log_table_bwd", l._table.data_ptr(), dtable.data_ptr(), self.grads[name].data_ptr(), F, K, Cn, 1, stream)
            elif isinstance(l, HipGaussianLayer):  # (lse-sum) batch sums of d log N / d mean, d log N / d stddev
                mean, stddev, _ = l._vals
                capi.call("ck_gaussian_bwd", g.data_ptr(), bd.xt.data_ptr(), l._scope(g.device).data_ptr(), mean.data_ptr(), stddev.data_ptr(),
                          sc["dm"].data_ptr(), sc["dsd"].data_ptr(), F, B, K, stream)
                l.mean.backward(sc["dm"].view(mean.shape), self.grads, stream)
                l.stddev.backward(sc["dsd"].view(stddev.shape), self.grads, stream)
            elif isinstance(l, HipEmbeddingLayer):
                # out = log(w[f, :, x]): the scatter-add of Re(gout) over the batch, divided by w
                Cn = l.num_states
                if l._table.is_complex():
                    raise NotImplementedError("squared-circuit training: complex Embedding weights (the layer-level autograd of "
                                              "cirkit_amd.layer_ops.embedding differentiates them)")
                if sc["one_launch"]:
                    capi.call("ck_embedding_bwd", g.data_ptr(), e, None, None, bd.xt_i.data_ptr(), l._scope(g.device).data_ptr(), l._table.data_ptr(),
                              sc["dw"].data_ptr(), F, B, K, Cn, stream)
                else:
                    gr = real_part(g, sc)
                    capi.call("ck_categorical_bwd", gr.data_ptr(), None, bd.xt_i.data_ptr(), l._scope(gr.device).data_ptr(),
                              sc["dtable"].data_ptr(), F, B, K, Cn, 0, None, stream)
                    capi.call("ck_embedding_weight_bwd", l._table.data_ptr(), sc["dtable"].data_ptr(), sc["dw"].data_ptr(), F, Cn, K, stream)
                l.weight.backward(sc["dw"], self.grads, stream)
            elif isinstance(l, HipConstantValueLayer):
                v = l._val
                if v.is_complex():
                    raise NotImplementedError("squared-circuit training: complex constant values")
                if B == 1:
                    if cplx:
                        capi.call("ck_copy_strided_f32", g.data_ptr(), sc["gsum"].data_ptr(), F * K, 2, 1, stream)
                        gsum = sc["gsum"]
                    else:
                        gsum = g.view(F, K)
                else:  # the batch sum as a product with a row of ones
                    gr = real_part(g, sc)
                    gsum = sc["gsum"]
                    capi.call("ck_param_bmm", sc["ones"].data_ptr(), gr.data_ptr(), gsum.data_ptr(), F, 1, K, B, 0, 0, 0, stream)
                if l.log_space:
                    dv = gsum
                else:
                    dv = sc["dv"]
                    capi.call("ck_param_unary_bwd", capi.CK_UNARY_LOG, v.data_ptr(), v.data_ptr(), gsum.data_ptr(), dv.data_ptr(), F * K, 0, stream)
                l.value.backward(dv.view(v.shape), self.grads, stream)
            else:  # (checked in __init__)
                raise NotImplementedError(type(l).__name__)


class _SignedCircuit:
    """c(x) of a squared circuit with real parameters on SIGNED-LOG blocks (cirkit_amd/csrc/ck_signed.hip): fp32 log|v| plus one
    sign word per row instead of the reference's (log|v|, 0 or pi) pairs -- half the bytes and half the contractions of the
    complex layers, Embedding outputs never stored (the first sum layer reads the weight table), weight gradients added
    straight into the flat gradient.  Qualifies: Embedding layers (weight = a tensor, 32 units) under CP-T / arity-1 sum
    layers of 32 inputs and 32 (or, not directly over an Embedding layer, 1 .. 4) outputs whose weights are tensors; every
    fold read once; one scalar output.  `why` says what does not fit (the trainer then keeps the complex launch lists)."""

    def __init__(self, circuit: HipCircuit, grads: Mapping[str, torch.Tensor]) -> None:
        self.c, self.grads = circuit, grads
        self.why = self._analyse()
        self._bound: dict[int, dict] = {}

    def _analyse(self) -> str | None:
        c = self.c
        if c.plan.semiring != "complex-lse-sum":
            return "not a complex-lse-sum circuit"
        if len(c._out_pairs) != 1:
            return "several outputs"
        self.kind: dict[int, str] = {}
        self.wname: dict[int, str] = {}
        for i, (spec, l) in enumerate(zip(c.plan.layers, c.layers)):
            ch = c._children[i]
            if isinstance(l, HipEmbeddingLayer):
                words = (l.num_states + 1) * 33 + 2 * l.num_states + 3 + 4096
                if l.num_output_units != 32 or l.weight.ops != ["tensor"] or words * 4 > 160 * 1024 or l.scope_idx.shape[1] != 1:
                    return f"layer {i}: Embedding layers need 32 units, a plain weight tensor and a table that fits the LDS"
                self.kind[i] = "emb"
                self.wname[i] = l.weight.graph.nodes[0].config["tensor"]
            elif type(l) in (HipSumLayer, HipCPTLayer) and not getattr(l, "_mixing", False):
                if type(l) is HipSumLayer and l.arity != 1:
                    return f"layer {i}: a sum layer over several concatenated children"
                Ko = l.num_output_units
                if l.num_input_units != 32 or not (Ko == 32 or 1 <= Ko <= 4) or l.weight.ops != ["tensor"]:
                    return f"layer {i}: needs 32 input units, 32 or 1 .. 4 output units and a plain weight tensor"
                kinds = {self.kind.get(int(p)) for p in np.unique(ch[..., 0])}
                if kinds == {"emb"}:
                    if Ko != 32 or len(np.unique(ch[..., 0])) != 1:
                        return f"layer {i}: a layer over Embedding folds needs 32 outputs and ONE Embedding layer beneath it"
                    self.kind[i] = "gather"
                elif kinds == {"sum"} or kinds == {"sum", "gather"} or kinds == {"gather"}:
                    self.kind[i] = "sum"
                else:
                    return f"layer {i}: children of mixed kinds"
                self.wname[i] = l.weight.graph.nodes[0].config["tensor"]
            else:
                return f"layer {i}: {spec.type!r} has no signed-log form"
        names = list(self.wname.values())
        if len(set(names)) != len(names):
            return "a parameter tensor shared between layers"
        for n in names:
            if self.c.store[n].is_complex():
                return "complex parameter tensors"
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        if self.kind.get(po) not in ("sum", "gather") or c.layers[po].num_output_units != 1:
            return "the output is not a scalar sum unit"
        import os

        # The LEAF REGION -- Embedding -> 2 or 4 levels of binary CP-T layers, every fold read once (fusion.find_subtree_groups) -- as
        # ONE forward launch on signed LINEAR tiles (ck_leaf_walk_fwd, signed + keep_levels: no exponential or logarithm below the
        # region's roots, the tiles of every second level kept) and one backward launch per two levels (ck_leaf_walk_bwd, is_signed):
        # what the trainer of real circuits does for Categorical -> CP-T regions (training.py), on values of either sign.
        self.leaf = self._leaf_region() if os.environ.get("CK_SLSE_LEAF", "1") != "0" else None
        for i, k in self.kind.items():  # an Embedding fold read by nobody would keep an unwritten gradient block
            if k == "emb":
                read = np.zeros(c.layers[i].num_folds, dtype=bool)
                for j, ch in enumerate(c._children):
                    if ch is not None:
                        read[ch[..., 1][ch[..., 0] == i]] = True
                if not read.all():
                    return f"layer {i}: Embedding folds that nothing reads"
        return None

    def _leaf_region(self):
        """The fused leaf region of c (a `fusion.SubtreeGroup` of depth 2 or 4), or None."""
        from .fusion import find_subtree_groups

        c = self.c
        for depth in (4, 2):
            groups = [g for g in find_subtree_groups(c.plan, c.layers, c._children, c._out_pairs, max_depth=depth, signed=True)
                      if g.depth == depth]
            if groups:
                break
        else:
            return None
        if len(groups) != 1:  # (several Embedding layers with a region each: layer by layer)
            return None
        g = groups[0]
        if (self.kind.get(g.input_layer) != "emb" or self.kind.get(g.levels[0]) != "gather"
                or any(self.kind.get(j) != "sum" for j in g.levels[1:]) or c.layers[g.input_layer].num_states >= 65535):
            return None
        return g

    def _bind_leaf(self, st: dict, B: int, Bp: int, parent: dict) -> dict:
        """Descriptors and buffers of the leaf region's launches at batch size B."""
        from .fusion import balanced_segments, leaf_segments

        c, g = self.c, self.leaf
        dev = c.device
        emb = c.layers[g.input_layer]
        D, kl, tiles = g.depth, 1 << g.depth, Bp // 32
        n_wg = c._n_cu
        n_roots = c.layers[g.root].num_folds
        nodes = np.asarray(g.nodes).astype(np.int64)
        noff = [int(v) for v in g.node_off]
        var_of_leaf = emb.scope_idx[:, 0].astype(np.int64)

        def lvl(l: int, t: int, j: int) -> int:  # fold of the j-th node of level l under root t (level 0: Embedding folds)
            return int(nodes[noff[l] + t * (kl >> l) + j])

        # where a root finds the gradient of its output: the block of the fold that reads it, as a block index of the gradient arena
        gin_block = np.asarray([(st["off"][parent[(g.root, f)][0]] + parent[(g.root, f)][1] * Bp * 32) // (Bp * 32) for f in range(n_roots)],
                               dtype=np.int32)
        launches = []  # top first: (unit table, level of P, work segments)
        for top in range(D, 0, -2):
            per_root = kl >> top
            tab = np.zeros((n_roots * per_root, 16), dtype=np.int32)
            for t in range(n_roots):
                for j in range(per_root):
                    r = tab[t * per_root + j]
                    r[0] = gin_block[lvl(D, t, 0)] if top == D else lvl(top + 1, t, j >> 1)
                    r[1] = lvl(top, t, j)
                    r[2], r[3] = lvl(top - 1, t, 2 * j), lvl(top - 1, t, 2 * j + 1)
                    for i in range(4):
                        r[4 + i] = lvl(top - 2, t, 4 * j + i)
                        if top == 2:
                            r[8 + i] = var_of_leaf[r[4 + i]]
                    r[12] = t
            work = balanced_segments(int(tab.shape[0]), tiles, n_wg, waves=8)
            launches.append((torch.from_numpy(tab).to(dev), top, torch.from_numpy(work).to(dev)))
        nodes_dev = torch.from_numpy(np.ascontiguousarray(g.nodes)).to(dev)
        scope = emb._scope(dev)
        node_off_c = (C.c_int32 * (D + 1))(*noff)
        keep = [torch.empty((c.layers[j].num_folds, tiles, 1024), dtype=torch.float32, device=dev) if l % 2 == 1 else None
                for l, j in enumerate(g.levels)]
        return {
            "launches": launches, "keep": keep,
            "redo": torch.zeros(n_roots * tiles, dtype=torch.int32, device=dev),
            "G": [torch.empty((c.layers[g.levels[top - 2]].num_folds, tiles, 1024), dtype=torch.float32, device=dev) if top > 2 else None
                  for _, top, _ in launches],
            "nodes": nodes_dev, "node_off": node_off_c, "scope": scope,
            "scale": torch.zeros((emb.num_folds, emb.num_states + 1), dtype=torch.float32, device=dev),
            "work": torch.from_numpy(leaf_segments(n_roots, tiles, n_wg)).to(dev), "n_wg": n_wg,
            "root_tab": c._leaf_root_table(nodes_dev, node_off_c, g.leaf_off, scope, D, n_roots),
            "x64": torch.zeros((B, max(1, c.plan.num_variables)), dtype=torch.int64, device=dev),
            "gin_fold": torch.from_numpy(gin_block[[lvl(D, t, 0) for t in range(n_roots)]].copy()).to(dev),
            "pairs": 1 if c._leaves_in_adjacent_pairs(g) else 0,
        }

    def bind(self, B: int) -> dict:
        st = self._bound.get(B)
        if st is not None:
            return st
        c = self.c
        dev = c.device
        off, n = {}, 0
        Bp = (B + 31) // 32 * 32  # rows of a fold's block: whole 32-row tiles (TILE-NATIVE blocks, ck_signed.hip)
        for i, k in self.kind.items():  # a block of (F, Bp, 32) floats per sum layer -- values, and at the same offset in the
            if k != "emb":              # gradient arena the gradient w.r.t. the product of its children
                off[i] = n
                n += c.layers[i].num_folds * Bp * 32  # (a 1 .. 4 unit layer: rows of Ko floats at the start of its block)
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        st = {
            "arena": torch.zeros(max(n, 32), dtype=torch.float32, device=dev),
            "signs": torch.zeros(max(n // 32, 1), dtype=torch.int32, device=dev),
            "garena": torch.zeros(max(n, 32), dtype=torch.float32, device=dev),
            "seed": torch.zeros(c.layers[po].num_folds * B, dtype=torch.float32, device=dev),  # the output layer's gradient
            "xt": torch.zeros((max(1, c.plan.num_variables), B), dtype=torch.int32, device=dev),
            "off": off, "ro": {}, "tabs": {}, "gout_off": {}, "gfold": {}, "ltab": {},
        }
        parent: dict[tuple[int, int], tuple[int, int]] = {}  # (layer, fold) -> the (layer, fold) that reads it
        for i, k in self.kind.items():
            if k == "emb":
                l = c.layers[i]
                rows = l.num_folds * (l.num_states + 1)
                st["ltab"][i] = (torch.zeros(rows * 32, dtype=torch.float32, device=dev), torch.zeros(rows, dtype=torch.int32, device=dev))
                continue
            ch = c._children[i]  # (F, H, 2)
            for f in range(ch.shape[0]):
                for h in range(ch.shape[1]):
                    parent[(int(ch[f, h, 0]), int(ch[f, h, 1]))] = (i, f)
            if k == "gather":
                emb = c.layers[int(ch[0, 0, 0])]
                folds = ch[..., 1].astype(np.int32)
                st["tabs"][i] = (torch.from_numpy(np.ascontiguousarray(folds)).to(dev),
                                 torch.from_numpy(np.ascontiguousarray(emb.scope_idx[folds, 0].astype(np.int32))).to(dev), int(ch[0, 0, 0]))
            else:
                ro = np.zeros(ch.shape[:2], dtype=np.int64)
                for p in np.unique(ch[..., 0]):
                    sel = ch[..., 0] == p
                    ro[sel] = off[int(p)] + ch[..., 1][sel].astype(np.int64) * Bp * 32
                st["ro"][i] = torch.from_numpy(ro).to(dev)
        for i, k in self.kind.items():  # where each fold finds the gradient of its output: its reader's block
            F = c.layers[i].num_folds
            if i == po:
                continue
            if k == "emb":
                gather = {parent[(i, f)][0] for f in range(F)}
                assert len(gather) == 1
                gfold = np.asarray([parent[(i, f)][1] for f in range(F)], dtype=np.int32)
                # folds that share a gradient block: 8 workgroups apart (one XCD, the same time: the block is read from HBM once)
                by_block: dict[int, list[int]] = {}
                for f in range(F):
                    by_block.setdefault(int(gfold[f]), []).append(f)
                groups, order = list(by_block.values()), []
                for i0 in range(0, len(groups), 8):
                    chunk = groups[i0 : i0 + 8]
                    for m in range(max(len(gr) for gr in chunk)):
                        order += [gr[m] if m < len(gr) else -1 for gr in chunk]
                if -1 in order or len(chunk) != 8:  # (ragged groups: the plain order)
                    order = list(range(F))
                assert sorted(order) == list(range(F))
                st["gfold"][i] = (torch.from_numpy(gfold).to(dev), gather.pop(), torch.from_numpy(np.asarray(order, dtype=np.int32)).to(dev))
            else:
                if any((i, f) not in parent for f in range(F)):
                    raise NotImplementedError(f"layer {i}: folds that nothing reads")
                st["gout_off"][i] = torch.from_numpy(np.asarray(
                    [off[parent[(i, f)][0]] + parent[(i, f)][1] * Bp * 32 for f in range(F)], dtype=np.int64)).to(dev)
        if self.leaf is not None:
            if B * max(1, c.plan.num_variables) * 8 >= 2**32 or B * 32 >= 2**31:
                raise NotImplementedError(f"squared-circuit training: a batch of {B} rows exceeds the 32-bit offsets of the leaf launches")
            st["leaf"] = self._bind_leaf(st, B, Bp, parent)
        while len(self._bound) >= 4:
            self._bound.pop(next(iter(self._bound)))
        self._bound[B] = st
        return st

    def output(self, B: int) -> torch.Tensor:
        """(B,) fp32: log|c(x_b)|."""
        c, st = self.c, self.bind(B)
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        o = st["off"][po] + fo * B
        return st["arena"][o : o + B]

    def stage(self, x: torch.Tensor, stream: int) -> None:
        """The (B, D) batch -> (D, B) int32, checked against the number of states (`HipCircuit._stage_input`)."""
        c = self.c
        B = int(x.shape[0])
        st = self.bind(B)
        _, xi = c._prepare_input(x)
        D = c.plan.num_variables
        x64 = st["leaf"]["x64"] if "leaf" in st else None  # (the leaf launches read the batch as it is: a copy at a recorded address)
        if c.validate_inputs:
            capi.call("ck_stage_categories", xi.data_ptr(), st["xt"].data_ptr(), B, D, c._num_states_dev().data_ptr(),
                      c._bad_input.data_ptr(), 1 if c._preclamp() else 0, None if x64 is None else x64.data_ptr(), stream)
        else:
            capi.call("ck_transpose_i64_to_i32", xi.data_ptr(), st["xt"].data_ptr(), B, D, stream)
            if x64 is not None:
                x64.copy_(xi)

    def _args(self, st: dict, i: int):
        c = self.c
        if self.kind[i] == "gather":
            folds, variables, e = st["tabs"][i]
            ltab, tsg = st["ltab"][e]
            return None, ltab.data_ptr(), tsg.data_ptr(), folds.data_ptr(), variables.data_ptr(), st["xt"].data_ptr(), c.layers[e].num_states
        return st["ro"][i].data_ptr(), None, None, None, None, None, 0

    def forward(self, B: int, stream: int) -> None:
        c, st = self.c, self.bind(B)
        a, sg = st["arena"].data_ptr(), st["signs"].data_ptr()
        for i, k in self.kind.items():
            l = c.layers[i]
            if k == "emb":  # the weight table (F, C + 1, 32) of this step's parameters, and its signed-log form
                if l._table is None or l._table.dtype != torch.float32:
                    l.prepare(stream, batched=False)  # (allocates the table; first call only)
                ltab, tsg = st["ltab"][i]
                in_region = self.leaf is not None and i == self.leaf.input_layer  # (the leaf launches read the linear table only)
                capi.call("ck_slse_tables", c.store[self.wname[i]].data_ptr(), l._table.data_ptr(), None if in_region else ltab.data_ptr(),
                          None if in_region else tsg.data_ptr(), l.num_folds, l.num_states, stream)
                continue
            if self.leaf is not None and i in self.leaf.levels:  # (the whole region with the launch of its first level)
                if i == self.leaf.levels[0]:
                    self._leaf_forward(st, B, stream)
                continue
            o = st["off"][i]
            ro, *gather = self._args(st, i)
            capi.call("ck_slse_fwd", a, sg, ro, c.store[self.wname[i]].data_ptr(), a + 4 * o, sg + 4 * (o // 32),
                      l.num_folds, l.arity, B, l.num_output_units, *gather, stream)
        if c.validate_inputs and c._int_input and self.leaf is None:  # (the leaf launch checks the rows it reads: theirs are NaN)
            y = self.output(B)
            capi.call("ck_poison_outputs", y.data_ptr(), B, c._bad_input.data_ptr(), stream)

    def _leaf_forward(self, st: dict, B: int, stream: int) -> None:
        c, g, L = self.c, self.leaf, st["leaf"]
        emb = c.layers[g.input_layer]
        o = st["off"][g.root]
        d = capi.LeafLaunch()
        d.table, d.table_scale, d.scope = emb._table.data_ptr(), L["scale"].data_ptr(), L["scope"].data_ptr()
        d.w_levels = (C.c_void_p * g.depth)(*[c.store[self.wname[j]].data_ptr() for j in g.levels])
        d.nodes, d.node_off, d.leaf_off = L["nodes"].data_ptr(), L["node_off"], g.leaf_off
        d.out, d.signs_out = st["arena"].data_ptr() + 4 * o, st["signs"].data_ptr() + 4 * (o // 32)
        d.work, d.n_seg, d.n_wg, d.waves, d.depth = L["work"].data_ptr(), int(L["work"].shape[0]), L["n_wg"], 8, g.depth
        d.B, d.K, d.C, d.w_layout = B, 32, emb.num_states, capi.CK_W_ROWMAJOR
        d.signed_redo = d.keep_redo = L["redo"].data_ptr()
        d.n_roots, d.root_tab = c.layers[g.root].num_folds, L["root_tab"].data_ptr()
        d.xt, d.preclamped, d.D, d.x_rows, d.x_input = None, 0, c.plan.num_variables, L["x64"].data_ptr(), -1
        d.bad_input = c._bad_input.data_ptr() if c.validate_inputs else None
        d.x_pairs = L["pairs"]
        d.keep_levels = (C.c_void_p * g.depth)(*[None if t is None else t.data_ptr() for t in L["keep"]])
        capi.call("ck_leaf_walk_fwd", C.byref(d), stream)

    def _leaf_backward(self, st: dict, B: int, stream: int) -> None:
        c, g, L = self.c, self.leaf, st["leaf"]
        emb = c.layers[g.input_layer]
        ga = st["garena"].data_ptr()
        gout1 = ga + 4 * st["off"][g.levels[0]]  # (F_1, B, 32) row-major: what ck_embedding_bwd scatters
        gin = ga
        for k, (tab, top, work) in enumerate(L["launches"]):
            lp, lq = g.levels[top - 1], g.levels[top - 2]
            d = capi.LeafBwdLaunch()
            d.unit_tab, d.work, d.n_seg, d.n_wg, d.B, d.waves = tab.data_ptr(), work.data_ptr(), int(work.shape[0]), L["n_wg"], B, 8
            d.C, d.D, d.leaf, d.is_signed = emb.num_states, c.plan.num_variables, 1 if top == 2 else 0, 1
            d.gin, d.gin_rowmajor = gin, 0
            d.y_p = L["keep"][top - 1].data_ptr()
            if top == 2:
                d.table, d.x_rows = emb._table.data_ptr(), L["x64"].data_ptr()
            else:
                d.y_c = L["keep"][top - 3].data_ptr()
            d.w_p, d.w_q = c.store[self.wname[lp]].data_ptr(), c.store[self.wname[lq]].data_ptr()
            d.dw_p, d.dw_q = self.grads[self.wname[lp]].data_ptr(), self.grads[self.wname[lq]].data_ptr()
            d.gout = gout1 if top == 2 else L["G"][k].data_ptr()
            d.redo = L["redo"].data_ptr()
            capi.call("ck_leaf_walk_bwd", C.byref(d), stream)
            gin = d.gout
        # (root, tile) units whose forward walk left the linear range: in signed log space, by a launch in which every other wave exits
        capi.call("ck_leaf_walk_bwd_redo", emb._table.data_ptr(), L["scale"].data_ptr(), L["x64"].data_ptr(), B, emb.num_states,
                  c.plan.num_variables, L["nodes"].data_ptr(), L["node_off"], g.leaf_off, L["scope"].data_ptr(), g.depth,
                  (C.c_void_p * g.depth)(*[c.store[self.wname[j]].data_ptr() for j in g.levels]),
                  (C.c_void_p * g.depth)(*[self.grads[self.wname[j]].data_ptr() for j in g.levels]),
                  ga, gout1, L["redo"].data_ptr(), c.layers[g.root].num_folds, L["gin_fold"].data_ptr(), 1, stream)

    def backward(self, B: int, seed: float, stream: int) -> None:
        """Gradients of ``seed * sum_b log|c(x_b)|`` ADDED into `grads` (the Embedding weights': written)."""
        c, st = self.c, self.bind(B)
        a, sg, ga = st["arena"].data_ptr(), st["signs"].data_ptr(), st["garena"].data_ptr()
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        if st.get("seed_key") != float(seed):  # (nobody writes this block: filled once per binding and seed, outside the recorded lists)
            if c.layers[po].num_folds > 1:
                capi.call("ck_fill_f32", st["seed"].data_ptr(), st["seed"].numel(), 0.0, stream)
            capi.call("ck_fill_f32", st["seed"].data_ptr() + 4 * fo * B, B, float(seed), stream)
            st["seed_key"] = float(seed)
        for i in reversed(list(self.kind)):
            l, k = c.layers[i], self.kind[i]
            if k == "emb":
                gfold, g, order = st["gfold"][i]
                capi.call("ck_embedding_bwd", ga + 4 * st["off"][g], 1, gfold.data_ptr(), order.data_ptr(), st["xt"].data_ptr(), l._scope(c.device).data_ptr(),
                          l._table.data_ptr(), self.grads[self.wname[i]].data_ptr(), l.num_folds, B, 32, l.num_states, stream)
                continue
            if self.leaf is not None and i in self.leaf.levels:  # (the whole region when its root layer is reached)
                if i == self.leaf.root:
                    self._leaf_backward(st, B, stream)
                continue
            o = st["off"][i]
            ro, *gather = self._args(st, i)
            gout, gout_off = (st["seed"].data_ptr(), None) if i == po else (ga, st["gout_off"][i].data_ptr())
            capi.call("ck_slse_bwd", a, sg, ro, c.store[self.wname[i]].data_ptr(), a + 4 * o, sg + 4 * (o // 32), gout, gout_off,
                      ga + 4 * o, self.grads[self.wname[i]].data_ptr(), l.num_folds, l.arity, B, l.num_output_units, *gather, stream)


class HipSquaredTrainer:
    """Maximum-likelihood training of a squared circuit with real parameters: ``loss = -mean_b (2 Re c(x_b) - Re Z)``
    (the reference's loop for sum-of-squares circuits; c under complex-lse-sum -- or a real circuit under lse-sum --, Z built
    from the plan of c)."""

    def __init__(self, plan_c: Plan, tensors: Mapping[str, object], *, plan_z: Plan | None = None, device: str | torch.device = "cuda:0",
                 lr: float = 0.01, optimizer: str = "adam", betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 use_graph: bool = False, signed: bool | None = None) -> None:
        if plan_c.semiring not in ("complex-lse-sum", "lse-sum"):
            raise NotImplementedError(f"HipSquaredTrainer: semiring {plan_c.semiring!r}")
        if optimizer not in ("adam", "sgd"):
            raise ValueError(f"unknown optimizer {optimizer!r}")
        if plan_z is None:
            from .functional import squared_partition_plan

            plan_z = squared_partition_plan(plan_c)
        dev = torch.device(device)
        names = list(plan_c.tensors)
        sizes = [int(np.prod(plan_c.tensors[n][0])) for n in names]
        self._flat_param = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        store = TensorStore(dev)
        off = 0
        for n, sz in zip(names, sizes):
            v = tensors[n]
            if np.iscomplexobj(v) or (hasattr(v, "is_complex") and v.is_complex()):
                raise NotImplementedError("HipSquaredTrainer: complex parameter tensors")
            view = self._flat_param[off : off + sz].view(plan_c.tensors[n][0])
            view.copy_(torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach().to(torch.float32))
            store._t[n] = view
            off += sz
        store.version += 1
        self.plan_c, self.plan_z, self.store = plan_c, plan_z, store
        # (every layer evaluates its own parameter graph -- no batched prologue --: `HipParameter.backward` differentiates what
        #  `HipParameter.evaluate` left behind)
        kw = dict(device=dev, use_graph=False, fuse=False, pad_units=False, signed_real=False, tiled_weights=False, dense_on_table=False,
                  fused_weight_softmax=False, batch_params=False)
        self.c = HipCircuit(plan_c, store, **kw)
        self.z = HipCircuit(plan_z, store, **kw)
        self.device = self.c.device
        self._flat_grad = torch.zeros(sum(sizes), dtype=torch.float32, device=self.device)
        self.grads: dict[str, torch.Tensor] = {}
        off = 0
        for n, sz in zip(names, sizes):
            self.grads[n] = self._flat_grad[off : off + sz].view(plan_c.tensors[n][0])
            off += sz
        # Z's launches run beside c's on a second stream: their gradients go to a buffer of their own, added before the optimizer
        self._flat_grad_z = torch.zeros_like(self._flat_grad)
        grads_z, off = {}, 0
        for n, sz in zip(names, sizes):
            grads_z[n] = self._flat_grad_z[off : off + sz].view(plan_c.tensors[n][0])
            off += sz
        self._bwd_c, self._bwd_z = _PlanBackward(self.c, self.grads), _PlanBackward(self.z, grads_z)
        # c on signed-log blocks where its layers allow it (`signed`: None = where they do, True = required, False = never)
        sc = _SignedCircuit(self.c, self.grads) if signed is not False else None
        if signed is True and sc.why is not None:
            raise NotImplementedError(f"HipSquaredTrainer(signed=True): {sc.why}")
        self._signed = sc if (sc is not None and sc.why is None) else None
        self.lr, self.optimizer, self.betas, self.eps = lr, optimizer, betas, eps
        self._m1 = torch.zeros_like(self._flat_grad) if optimizer == "adam" else None
        self._m2 = torch.zeros_like(self._flat_grad) if optimizer == "adam" else None
        self._bad_seen = torch.zeros(1, dtype=torch.int32, device=self.device)  # latched by `step`, reported by `check_inputs`
        self._ll = torch.zeros(2, dtype=torch.float64, device=self.device)
        self._opt: torch.Tensor | None = None  # the DEVICE ck_opt_state: constants, clock, dropped steps
        self._opt_key = None
        self.use_graph = bool(use_graph)
        self._programs: dict[tuple, tuple] = {}
        self._side: tuple[torch.cuda.Stream, torch.cuda.Stream] | None = None

    def __del__(self):
        try:
            for pr, _ in self._programs.values():
                capi.load().ck_program_destroy(pr)
        except Exception:
            pass

    # -- the recorded step ----------------------------------------------------------------------------------------------
    def _opt_state(self) -> torch.Tensor:
        key = (float(self.lr), tuple(float(b) for b in self.betas), float(self.eps))
        if self._opt is None:
            o = capi.OptState()
            o.lr, o.b1, o.b2, o.eps, o.bc1, o.bc2 = self.lr, self.betas[0], self.betas[1], self.eps, 1.0, 1.0
            o.step, o.skipped, o.skip_now, o.kind = 0, 0, 0, 1 if self.optimizer == "adam" else 0
            o.b1d, o.b2d = float(self.betas[0]), float(self.betas[1])  # the bias corrections are formed in double (torch.optim.Adam does)
            self._opt = torch.frombuffer(bytearray(bytes(o)), dtype=torch.uint8).to(self.device)
        elif key != self._opt_key:  # (the learning rate was changed between steps: the first 16 bytes)
            head = torch.tensor([self.lr, self.betas[0], self.betas[1], self.eps], dtype=torch.float32).view(torch.uint8)
            self._opt[:16].copy_(head.to(self.device))
            self._opt[40:56].copy_(torch.tensor([self.betas[0], self.betas[1]], dtype=torch.float64).view(torch.uint8).to(self.device))
        self._opt_key = key
        return self._opt

    def _enqueue_optimizer(self, stream: int, with_z: bool = False) -> None:
        """`with_z`: the optimizer reads c's and Z's gradient buffers and adds them itself (no axpy launch before it; `grads`
        then holds c's part only -- `loss_and_grads` leaves the sum)."""
        p = self._flat_param
        capi.call("ck_opt_step_range", p.data_ptr(), self._flat_grad.data_ptr(), self._flat_grad_z.data_ptr() if with_z else None,
                  None if self._m1 is None else self._m1.data_ptr(), None if self._m2 is None else self._m2.data_ptr(), p.numel(),
                  self._opt_state().data_ptr(), stream)

    def _enqueue(self, part: str, B: int, gB: float, with_optimizer: bool, stream: int) -> None:
        n = self._flat_grad.numel()
        if part == "pre":  # every buffer the two backward lists ADD into, zeroed on c's stream: Z's list is the longer chain beside
            # c's whole-chip launches (LAB_NOTES R6.4), its backward waits for this list behind its forward
            capi.call("ck_fill_f32", self._flat_grad.data_ptr(), n, 0.0, stream)
            capi.call("ck_fill_f32", self._flat_grad_z.data_ptr(), n, 0.0, stream)
            self._bwd_z.zero_pool(1, stream)
        elif part == "c":
            if self._signed is not None:  # (its forward is part of the list: only the staging of the batch is not)
                self._signed.forward(B, stream)
                self._signed.backward(B, -2.0 / gB, stream)
            else:
                self._bwd_c.run(B, -2.0 / gB, stream)
        elif part == "zf":  # parameters of Z and its forward (no input: everything is part of the list) ...
            self.z._enqueue_params(stream)
            self._bwd_z.forward(1, stream)
        elif part == "zb":  # ... its backward
            self._bwd_z.run(1, B / gB, stream, pool_zeroed=True)
        elif part == "mid":  # both forwards are there: the optimizer's clock and the log-likelihood pair (beside Z's backward)
            c, z = self.c, self.z
            validate = c.validate_inputs and c._int_input
            if with_optimizer:  # a batch with an illegal category drops the step (skip_now)
                capi.call("ck_opt_tick", self._opt_state().data_ptr(), c._bad_input.data_ptr() if validate else None,
                          self._bad_seen.data_ptr() if validate else None, stream)
            yc = self._signed.output(B) if self._signed is not None else c._bind(B).views[int(c._out_pairs[0, 0])][int(c._out_pairs[0, 1])]
            yz = z._bind(1).views[int(z._out_pairs[0, 0])][int(z._out_pairs[0, 1])]
            capi.call("ck_squared_ll", yc.data_ptr(), B, 2 if yc.is_complex() else 1, yz.data_ptr(), self._ll.data_ptr(), stream)
        else:  # both gradients are there: their sum, or the optimizer reading both
            if with_optimizer:
                self._enqueue_optimizer(stream, with_z=True)
            else:
                capi.call("ck_axpy_f32", self._flat_grad.data_ptr(), self._flat_grad_z.data_ptr(), 1.0, n, stream)

    def _part(self, part: str, B: int, gB: float, with_optimizer: bool, run: torch.cuda.Stream) -> None:
        """One of the recorded launch lists of a step -- "pre": the zero fills of both gradient buffers; "c": forward and backward of
        c; "zf" / "zb": parameters + forward and the backward of Z; "end": the sum of the two gradients, the log-likelihood pair and
        (alone) the optimizer ("mid": the clock and the pair, as soon as both forwards are there) -- per (batch size, global batch): the first two
        calls run eagerly (they size the scratch of the parameter graphs), the third records, later ones replay (as a hipGraph
        when `use_graph`)."""
        c_arena = self._signed.bind(B)["arena"] if self._signed is not None else self.c._bind(B).arena
        key = (part, B, float(gB), bool(with_optimizer), c_arena.data_ptr(), self.z._bind(1).arena.data_ptr())
        prog, seen = self._programs.get(key, (None, 0))
        if prog is None:
            if seen < 2:
                self._enqueue(part, B, gB, with_optimizer, run.cuda_stream)
                self._programs[key] = (None, seen + 1)
                return
            for k in [k for k in self._programs if k[:4] == key[:4] and k != key]:  # (rebound arenas: their lists are stale)
                if self._programs[k][0] is not None:
                    capi.load().ck_program_destroy(self._programs[k][0])
                del self._programs[k]
            prog = C.c_void_p()
            capi.call("ck_program_begin", C.byref(prog))
            try:
                self._enqueue(part, B, gB, with_optimizer, run.cuda_stream)
            finally:
                capi.call("ck_program_end", prog)
            self._programs[key] = (prog, seen)
        capi.call("ck_program_launch", prog, 1 if self.use_graph else 0, run.cuda_stream)

    def _launch(self, x: torch.Tensor, B: int, gB: float, with_optimizer: bool) -> None:
        """Forward and backward of c on the caller's stream, forward and backward of Z (a few hundred launches on one row: they
        fill a fraction of the chip) on a second stream beside it, then the end of the step."""
        cur = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))
        main = self._side[0] if (self.use_graph and cur.cuda_stream == 0) else cur  # (a capture cannot run on the legacy stream)
        side = self._side[1]
        if main is not cur:
            main.wait_stream(cur)
        side.wait_stream(cur)
        with torch.cuda.stream(main):  # (the long list first)
            self._part("pre", B, gB, with_optimizer, main)
            zeroed = torch.cuda.Event()
            zeroed.record(main)
            if self._signed is not None:
                self._signed.stage(x, main.cuda_stream)
            else:
                self.c._run(x)  # (B, 1, 1) complex64 / fp32 in c's arena
            self._part("c", B, gB, with_optimizer, main)
        self._part("zf", B, gB, with_optimizer, side)
        z_forward = torch.cuda.Event()
        z_forward.record(side)
        side.wait_event(zeroed)
        self._part("zb", B, gB, with_optimizer, side)
        main.wait_event(z_forward)
        self._part("mid", B, gB, with_optimizer, main)
        main.wait_stream(side)
        self._part("end", B, gB, with_optimizer, main)
        if main is not cur:
            cur.wait_stream(main)

    def _global_batch(self, B: int, global_batch: int | None) -> float:
        import torch.distributed as dist

        if global_batch is None and _world_size() > 1:
            global_batch = B * _world_size()
        return float(global_batch or B)

    def loss_and_grads(self, x: torch.Tensor, *, global_batch: int | None = None) -> torch.Tensor:
        """Forward of c on the batch and of Z, then both backward launch lists: the gradients of
        ``-(1 / global_batch) sum_b (2 Re c(x_b)) + (B / global_batch) Re Z`` land in `self.grads`; returns the device tensor
        ``[sum_b 2 Re c(x_b) - B Re Z, B]`` (the shard's summed log-likelihood and its rows; fp64, rewritten by the next call)."""
        with torch.cuda.device(self.device):
            B = int(x.shape[0])
            self._launch(x, B, self._global_batch(B, global_batch), False)
            return self._ll

    def all_reduce_grads(self) -> None:
        import torch.distributed as dist

        # RCCL through the C ABI (ck_comm_all_reduce_f32, on the launch stream) when a HipComm is set; torch.distributed otherwise
        if _default_comm() is not None or (dist.is_available() and dist.is_initialized()):
            _all_reduce_sum(self._flat_grad)

    def apply_gradients(self, skip_flag: torch.Tensor | None = None) -> None:
        """The optimizer step on `self.grads`.  `skip_flag`: a device int32; nonzero at launch time = the step changes nothing
        (parameters, moments, Adam's step count), the flag is latched into what `check_inputs()` reports and cleared."""
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            capi.call("ck_opt_tick", self._opt_state().data_ptr(), None if skip_flag is None else skip_flag.data_ptr(),
                      None if skip_flag is None else self._bad_seen.data_ptr(), stream)
            self._enqueue_optimizer(stream)
            self.store.touch()

    def step(self, x: torch.Tensor, *, global_batch: int | None = None) -> torch.Tensor:
        """One optimisation step.  A batch with an out-of-range category (an ``IndexError`` in the reference, NaN outputs here)
        must not reach the parameters, exactly as in `HipTrainer.step`: alone, the optimizer's clock (`ck_opt_tick`) moves the
        circuit's flag into the step's skip state and the update changes nothing; with several ranks this rank's gradients are
        zeroed before the all-reduce (every rank takes the same step).  The flag is latched into what `check_inputs()` reports
        and cleared -- no host synchronisation, and alone everything after the two forwards is one replayed launch list."""
        import torch.distributed as dist

        c = self.c
        validate = c.validate_inputs and c._int_input
        alone = _world_size() <= 1
        with torch.cuda.device(self.device):
            B = int(x.shape[0])
            self._launch(x, B, self._global_batch(B, global_batch), alone)
            if alone:
                self.store.touch()
                return self._ll
            stream = torch.cuda.current_stream(self.device).cuda_stream
            if validate:
                capi.call("ck_zero_if_flag", self._flat_grad.data_ptr(), self._flat_grad.numel(), c._bad_input.data_ptr(), stream)
        self.all_reduce_grads()
        self.apply_gradients(None)
        if validate:
            with torch.cuda.device(self.device):
                capi.call("ck_latch_flag", c._bad_input.data_ptr(), self._bad_seen.data_ptr(),
                          torch.cuda.current_stream(self.device).cuda_stream)
        return self._ll

    def opt_counters(self) -> tuple[int, int]:
        """(steps taken, steps dropped) of the optimizer's device clock (a device read)."""
        if self._opt is None:
            return 0, 0
        v = self._opt[24:32].cpu().view(torch.int32)
        return int(v[0]), int(v[1])

    @property
    def step_count(self) -> int:
        return self.opt_counters()[0]

    @property
    def skipped_steps(self) -> int:
        return self.opt_counters()[1]

    def check_inputs(self) -> None:
        """Raise ``IndexError`` if a batch since the last check held a category out of range (layers/input.py:258-266,
        399-412 index with it); on a single rank the steps on such batches changed nothing."""
        if int(self._bad_seen.item()) != 0:
            self._bad_seen.zero_()
            self.c._bad_input.zero_()
            raise IndexError("a batch held a category outside [0, num_categories) of its variable")
        self.c.check_inputs()

    def gradients(self) -> dict[str, np.ndarray]:
        return {n: g.detach().cpu().numpy() for n, g in self.grads.items()}

    def parameters(self) -> dict[str, np.ndarray]:
        return {n: self.store.export(n) for n in self.plan_c.tensors}
