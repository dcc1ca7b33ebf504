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

torch appears as storage and for data movement only (real parts, permutations of a handful of small tensors); every arithmetic
step is a kernel of the C ABI.  Restrictions (checked, `NotImplementedError`): real parameter tensors, every (layer, fold)
read by exactly one consumer (trees: what `squared_partition_plan` and the region-graph templates give), a scalar output."""
from __future__ import annotations

import ctypes as C
from typing import Mapping

import numpy as np
import torch

from . import _capi as capi
from .circuit import HipCircuit
from .layers import (HipCategoricalLayer, HipConstantValueLayer, HipCPTLayer, HipEmbeddingLayer, HipGaussianLayer, HipHadamardLayer,
                     HipSumLayer, HipTensorDotLayer, HipTuckerLayer)
from .parameters import TensorStore
from .plan import Plan


class _PlanBackward:
    """The reverse launch list of ONE layer-wise circuit (complex-lse-sum or lse-sum) over the activations of its last forward."""

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

    def _bind(self, B: int) -> dict:
        bd = self.c._bind(B)
        st = self._bound.get(B)
        if st is not None and st["arena_ptr"] == bd.arena.data_ptr():
            return st
        garena = torch.zeros_like(bd.arena)  # complex64 / fp32, the mirror of the activation arena
        gviews = []
        for i, l in enumerate(self.c.layers):
            off = (bd.views[i].data_ptr() - bd.arena.data_ptr()) // bd.arena.element_size()
            gviews.append(garena[off : off + l.num_folds * B * l.num_output_units].view(l.num_folds, B, l.num_output_units))
        # host copies of the child offsets (read once: a `.tolist()` per step would be a device synchronisation per layer), the
        # offset tables in float units for the Hadamard launches, and the row tables of the TensorDot layers' permuted inputs
        host_ro, ro2, td_rows = {}, {}, {}
        for i, l in enumerate(self.c.layers):
            if bd.row_off[i] is None:
                continue
            if isinstance(l, HipTensorDotLayer):
                host_ro[i] = bd.row_off[i].reshape(-1).cpu().numpy()
                n = B * l.num_input_units
                td_rows[i] = (torch.arange(l.num_folds, dtype=torch.int64, device=garena.device) * n).reshape(l.num_folds, 1)
            elif isinstance(l, HipHadamardLayer):
                ro2[i] = (bd.row_off[i] * (2 if self.cplx else 1)).contiguous()
        st = {"arena_ptr": bd.arena.data_ptr(), "garena": garena, "gviews": gviews, "host_ro": host_ro, "ro2": ro2, "td_rows": td_rows}
        while len(self._bound) >= 4:
            self._bound.pop(next(iter(self._bound)))
        self._bound[B] = st
        return st

    def run(self, B: int, seed_real: float, stream: int) -> None:
        """Gradients of ``seed_real * sum_b Re out_b`` w.r.t. the parameter tensors, ADDED into `grads`."""
        c = self.c
        bd = c._bind(B)
        st = self._bind(B)
        garena, gviews = st["garena"], st["gviews"]
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        gviews[po].zero_()
        gviews[po][fo] = complex(seed_real, 0.0) if self.cplx else seed_real
        ga = garena.data_ptr()
        cplx = self.cplx

        def sum_bwd(arena_ptr, garena_ptr, row_off, w, out_ptr, g_ptr, dw, F, H, rows, Ki, Ko, mode):
            if cplx:
                capi.call("ck_sum_lse_bwd_c", arena_ptr, garena_ptr, row_off.data_ptr(), w.data_ptr(), out_ptr, g_ptr, dw.data_ptr(),
                          F, H, rows, Ki, Ko, mode, 1 if w.is_complex() else 0, stream)
            else:
                capi.call("ck_sum_lse_bwd", arena_ptr, garena_ptr, row_off.data_ptr(), None, w.data_ptr(), out_ptr, g_ptr, dw.data_ptr(),
                          F, H, rows, Ki, Ko, mode, 0, stream)

        def real_part(g):
            return torch.view_as_real(g)[..., 0] if cplx else g

        for i in range(len(c.layers) - 1, -1, -1):
            l = c.layers[i]
            F, K = l.num_folds, l.num_output_units
            g = gviews[i]
            if isinstance(l, HipTensorDotLayer):
                Kj, Kq = l._num_contract_units, l._num_batch_units
                Kk = K // Kq
                ro = st["host_ro"][i]  # (arity 1: one (B, Kj * Kq) block per fold)
                n = B * Kj * Kq
                packed = bool(np.array_equal(ro, ro[0] + n * np.arange(F)))  # the producer's folds in order: one slice of the arena
                x = (bd.arena[int(ro[0]) : int(ro[0]) + F * n] if packed else
                     torch.stack([bd.arena[int(o) : int(o) + n] for o in ro])).view(F, B, Kj, Kq)
                # the layer IS a dense sum over the rows (b, q) of the permuted input (optimized.py:289-296)
                xp = x.permute(0, 1, 3, 2).contiguous().view(F, B * Kq, Kj)
                gx = torch.empty_like(xp)
                w = l._w
                if w.is_complex():
                    raise NotImplementedError("squared-circuit training: complex-valued weights")
                dw = torch.zeros((F, Kk, Kj), dtype=torch.float32, device=w.device)
                rows = st["td_rows"][i]
                sum_bwd(xp.data_ptr(), gx.data_ptr(), rows, w, bd.views[i].data_ptr(), g.data_ptr(), dw, F, 1, B * Kq, Kj, Kk, capi.CK_SUM_PROD)
                gxp = gx.view(F, B, Kq, Kj).permute(0, 1, 3, 2).contiguous().view(F, n)
                if packed:
                    garena[int(ro[0]) : int(ro[0]) + F * n] = gxp.reshape(-1)
                else:
                    for f, o in enumerate(ro):
                        garena[int(o) : int(o) + n] = gxp[f]
                l.weight.backward(dw, self.grads, stream)
            elif isinstance(l, (HipSumLayer, HipCPTLayer, HipTuckerLayer)):
                w = l._w
                if w.is_complex():
                    raise NotImplementedError("squared-circuit training: complex-valued weights")
                dw = torch.zeros_like(w)
                sum_bwd(bd.arena.data_ptr(), ga, bd.row_off[i], w, bd.views[i].data_ptr(), g.data_ptr(), dw, F, l.arity, B, l.num_input_units, K,
                        l._mode)
                l.weight.backward(dw, self.grads, stream)
            elif isinstance(l, HipHadamardLayer):  # log space: the sum of the children -- (re, im) pairs as 2 K floats
                e = 2 if cplx else 1
                capi.call("ck_hadamard_bwd", ga, st["ro2"][i].data_ptr(), g.data_ptr(), F, l.arity, B, e * K, 0, stream)
            elif isinstance(l, HipCategoricalLayer):  # (lse-sum) the scatter-add into the log-table, then log softmax backward
                Cn = l.num_categories
                dtable = torch.zeros((F, Cn + 1, K), dtype=torch.float32, device=g.device)
                capi.call("ck_categorical_bwd", g.data_ptr(), None, bd.xt_i.data_ptr(), l._scope(g.device).data_ptr(), dtable.data_ptr(),
                          F, B, K, Cn, 1, None, stream)
                name = l.probs.graph.nodes[0].config["tensor"]
                capi.call("ck_param_log_table_bwd", l._table.data_ptr(), dtable.data_ptr(), self.grads[name].data_ptr(), F, K, Cn, 1, stream)
            elif isinstance(l, HipGaussianLayer):  # (lse-sum) batch sums of d log N / d mean, d log N / d stddev
                mean, stddev, _ = l._vals
                dm, dsd = torch.empty_like(mean), torch.empty_like(stddev)
                capi.call("ck_gaussian_bwd", g.data_ptr(), bd.xt.data_ptr(), l._scope(g.device).data_ptr(), mean.data_ptr(), stddev.data_ptr(),
                          dm.data_ptr(), dsd.data_ptr(), F, B, K, stream)
                l.mean.backward(dm, self.grads, stream)
                l.stddev.backward(dsd, self.grads, stream)
            elif isinstance(l, HipEmbeddingLayer):
                # out = log(w[f, :, x]): the scatter-add of Re(gout) over the batch, divided by w
                Cn = l.num_states
                if l._table.is_complex():
                    raise NotImplementedError("squared-circuit training: complex Embedding weights (the layer-level autograd of "
                                              "cirkit_amd.layer_ops.embedding differentiates them)")
                gr = real_part(g).contiguous()
                dtable = torch.zeros((F, Cn + 1, K), dtype=torch.float32, device=gr.device)
                capi.call("ck_categorical_bwd", gr.data_ptr(), None, bd.xt_i.data_ptr(), l._scope(gr.device).data_ptr(), dtable.data_ptr(),
                          F, B, K, Cn, 1, None, stream)
                wt = l._table[:, :Cn].contiguous()  # (F, C, K): the weight, transposed like every gather table
                num = dtable[:, :Cn].contiguous()
                dwt = torch.empty_like(num)
                capi.call("ck_param_unary_bwd", capi.CK_UNARY_LOG, wt.data_ptr(), wt.data_ptr(), num.data_ptr(), dwt.data_ptr(), num.numel(), 0, stream)
                l.weight.backward(dwt.transpose(1, 2).contiguous(), self.grads, stream)
            elif isinstance(l, HipConstantValueLayer):
                v = l._val
                if v.is_complex():
                    raise NotImplementedError("squared-circuit training: complex constant values")
                gr = real_part(g)
                gsum = (gr[:, 0] if B == 1 else gr.sum(dim=1)).contiguous()  # (F, K)
                if l.log_space:
                    dv = gsum
                else:
                    dv = torch.empty_like(gsum)
                    capi.call("ck_param_unary_bwd", capi.CK_UNARY_LOG, v.data_ptr(), v.data_ptr(), gsum.data_ptr(), dv.data_ptr(), gsum.numel(), 0, stream)
                l.value.backward(dv.view(v.shape), self.grads, stream)
            else:  # (checked in __init__)
                raise NotImplementedError(type(l).__name__)


class HipSquaredTrainer:
    """Maximum-likelihood training of a squared circuit with real parameters: ``loss = -mean_b (2 Re c(x_b) - Re Z)``
    (the reference's loop for sum-of-squares circuits; c under complex-lse-sum -- or a real circuit under lse-sum --, Z built
    from the plan of c)."""

    def __init__(self, plan_c: Plan, tensors: Mapping[str, object], *, plan_z: Plan | None = None, device: str | torch.device = "cuda:0",
                 lr: float = 0.01, optimizer: str = "adam", betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8) -> None:
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
        self._bwd_c, self._bwd_z = _PlanBackward(self.c, self.grads), _PlanBackward(self.z, self.grads)
        self.lr, self.optimizer, self.betas, self.eps = lr, optimizer, betas, eps
        self._m1 = torch.zeros_like(self._flat_grad) if optimizer == "adam" else None
        self._m2 = torch.zeros_like(self._flat_grad) if optimizer == "adam" else None
        self._skipped = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._bad_seen = torch.zeros(1, dtype=torch.int32, device=self.device)  # latched by `step`, reported by `check_inputs`
        self.step_count = 0

    def loss_and_grads(self, x: torch.Tensor, *, global_batch: int | None = None) -> torch.Tensor:
        """Forward of c on the batch and of Z, then both backward launch lists: the gradients of
        ``-(1 / global_batch) sum_b (2 Re c(x_b)) + (B / global_batch) Re Z`` land in `self.grads`; returns the device tensor
        ``[sum_b 2 Re c(x_b) - B Re Z, B]`` (the shard's summed log-likelihood and its rows)."""
        import torch.distributed as dist

        with torch.cuda.device(self.device):
            B = int(x.shape[0])
            if global_batch is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                global_batch = B * dist.get_world_size()
            gB = float(global_batch or B)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            yc = self.c(x)          # (B, 1, 1) complex64 / fp32
            yz = self.z()           # (1, 1, 1)
            if not yc.is_complex():
                yc, yz = torch.complex(yc, torch.zeros_like(yc)), torch.complex(yz, torch.zeros_like(yz))
            capi.call("ck_fill_f32", self._flat_grad.data_ptr(), self._flat_grad.numel(), 0.0, stream)
            self._bwd_c.run(B, -2.0 / gB, stream)
            self._bwd_z.run(1, B / gB, stream)
            ll = 2.0 * yc.real.sum(dtype=torch.float64) - B * yz.real.reshape(()).to(torch.float64)
            return torch.stack([ll, torch.tensor(float(B), dtype=torch.float64, device=self.device)])

    def all_reduce_grads(self) -> None:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)

    def apply_gradients(self, skip_flag: torch.Tensor | None = None) -> None:
        """The optimizer step on `self.grads`.  `skip_flag`: a device int32; nonzero at launch time = the step changes nothing
        (parameters, moments, Adam's step count)."""
        with torch.cuda.device(self.device):
            self.step_count += 1
            stream = torch.cuda.current_stream(self.device).cuda_stream
            p, g = self._flat_param, self._flat_grad
            skip = None if skip_flag is None else skip_flag.data_ptr()
            if self.optimizer == "adam":
                capi.call("ck_adam_step", p.data_ptr(), g.data_ptr(), self._m1.data_ptr(), self._m2.data_ptr(), p.numel(), self.lr,
                          self.betas[0], self.betas[1], self.eps, self.step_count, 1.0, skip, self._skipped.data_ptr(), stream)
            else:
                capi.call("ck_sgd_step", p.data_ptr(), g.data_ptr(), p.numel(), self.lr, 1.0, skip, stream)
            self.store.touch()

    def step(self, x: torch.Tensor, *, global_batch: int | None = None) -> torch.Tensor:
        """One optimisation step.  A batch with an out-of-range category (an ``IndexError`` in the reference, NaN outputs here)
        must not reach the parameters, exactly as in `HipTrainer.step`: alone, the optimizer launch changes nothing while the
        circuit's flag is up; with several ranks this rank's gradients are zeroed before the all-reduce (every rank takes the
        same step).  The flag is then latched into what `check_inputs()` reports and cleared -- no host synchronisation."""
        import torch.distributed as dist

        ll = self.loss_and_grads(x, global_batch=global_batch)
        c = self.c
        validate = c.validate_inputs and c._int_input
        alone = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        if validate and not alone:
            with torch.cuda.device(self.device):
                capi.call("ck_zero_if_flag", self._flat_grad.data_ptr(), self._flat_grad.numel(), c._bad_input.data_ptr(),
                          torch.cuda.current_stream(self.device).cuda_stream)
        self.all_reduce_grads()
        self.apply_gradients(c._bad_input if (validate and alone) else None)
        if validate:
            with torch.cuda.device(self.device):
                capi.call("ck_latch_flag", c._bad_input.data_ptr(), self._bad_seen.data_ptr(),
                          torch.cuda.current_stream(self.device).cuda_stream)
        return ll

    @property
    def skipped_steps(self) -> int:
        return int(self._skipped.item())

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
