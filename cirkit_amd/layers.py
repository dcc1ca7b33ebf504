"""Host mirror of the reference's layer operator surface (``cirkit.backend.torch.layers``).

Every class keeps the constructor arguments, ``config`` / ``params`` / ``fold_settings`` properties
and the ``forward`` tensor contract of its reference counterpart (cited per class), so that the
folding machinery of the reference -- which re-instantiates ``type(layer)(semiring=..., **config,
num_folds=..., **params)`` (cirkit/backend/torch/compiler.py:374-406) -- and the ``module_fn`` hook
(graph/modules.py:224-237) can drive them.  The arithmetic is NOT here: ``forward`` only enqueues
HIP kernels through the C ABI (include/cirkit_hip.h).  torch tensors are storage.

Two entry styles per layer:

* ``forward(x)``: reference contract, ``x`` is the materialised ``(F, H, B, Ki)`` input (or
  ``(F, B, D')`` batch slice for input layers, or a batch size for constant layers).  The tensor is
  treated as a private arena with ``row_off[f, h] = (f*H + h) * B * Ki``.
* ``launch(rt, ...)``: used by `HipCircuit`, children addressed inside the shared activation arena
  (no gather copy).
"""

from __future__ import annotations

from typing import Any, Mapping

import numpy as np
import torch

from . import _capi as capi
from .parameters import HipParameter

SEMIRINGS = ("lse-sum", "complex-lse-sum")


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class HipLayer:
    """Counterpart of ``TorchLayer`` (layers/base.py:26-119)."""

    def __init__(
        self,
        num_input_units: int,
        num_output_units: int,
        arity: int = 1,
        *,
        semiring: str | None = None,
        num_folds: int = 1,
    ) -> None:
        if num_input_units < 0:
            raise ValueError("The number of input units must be non-negative")
        if num_output_units <= 0:
            raise ValueError("The number of output units must be positive")
        if arity <= 0:
            raise ValueError("The arity must be positive")
        semiring = semiring or "lse-sum"
        if semiring not in SEMIRINGS:
            raise ValueError(
                f"semiring {semiring!r} is not evaluated by the HIP backend (supported: {SEMIRINGS})"
            )
        self.num_input_units = num_input_units
        self.num_output_units = num_output_units
        self.arity = arity
        self.semiring = semiring
        self.num_folds = num_folds

    # -- reference surface -----------------------------------------------------------------------
    @property
    def config(self) -> Mapping[str, Any]:
        raise NotImplementedError

    @property
    def params(self) -> Mapping[str, HipParameter]:
        return {}

    @property
    def fold_settings(self) -> tuple[Any, ...]:
        pshapes = {n: p.shape for n, p in self.params.items()}
        return (*self.config.items(), *pshapes.items())

    @property
    def is_complex(self) -> bool:
        return self.semiring == "complex-lse-sum"

    @property
    def esize(self) -> int:
        """4-byte words per activation element."""
        return 2 if self.is_complex else 1

    @property
    def act_dtype(self) -> torch.dtype:
        return torch.complex64 if self.is_complex else torch.float32

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    # -- HIP side ----------------------------------------------------------------------------------
    def prepare(self, stream: int, batched: bool = False) -> None:
        """Enqueue the parameter-graph evaluation (and any layer-specific re-layout).  With
        ``batched`` the caller has already launched the `ParamBatch` this layer registered in."""

    def register_batched(self, batch) -> bool:
        """Hand the layer's softmax parameters to a circuit-wide `ParamBatch`; returns True if the
        layer needs no parameter work of its own afterwards."""
        return False

    _batched = False

    def _check_param(self, name: str, p: HipParameter, shape: tuple[int, ...]) -> None:
        if p.num_folds != self.num_folds or tuple(p.shape) != tuple(shape):
            raise ValueError(
                f"Expected number of folds {self.num_folds} and shape {tuple(shape)} for '{name}', "
                f"found {p.num_folds} and {tuple(p.shape)}, respectively"
            )


# =============================================================================================
# input layers  (layers/input.py)
# =============================================================================================
class HipInputLayer(HipLayer):
    """``TorchInputLayer`` (layers/input.py:13-123): owns ``scope_idx`` of shape (F, D')."""

    def __init__(self, scope_idx, num_output_units: int, *, semiring: str | None = None) -> None:
        scope_idx = np.asarray(
            scope_idx.detach().cpu().numpy() if hasattr(scope_idx, "detach") else scope_idx, dtype=np.int64
        )
        if scope_idx.ndim == 1:
            scope_idx = scope_idx[None]
        elif scope_idx.ndim > 2:
            raise ValueError(f"The scope index must be a matrix, but found shape {scope_idx.shape}")
        num_folds, num_variables = scope_idx.shape
        super().__init__(num_variables, num_output_units, semiring=semiring, num_folds=num_folds)
        self.scope_idx = scope_idx
        self._scope_dev: torch.Tensor | None = None

    @property
    def num_variables(self) -> int:
        return self.num_input_units

    def _scope(self, device) -> torch.Tensor:
        if self._scope_dev is None or self._scope_dev.device != device:
            if self.scope_idx.shape[1] != 1:
                # (F, D') is the base-class shape (input.py:13-123); every concrete input layer on this path refuses
                # D' != 1 in its constructor, in the reference (input.py:220-221, 345-346, 478-479, 603-604) and here
                raise NotImplementedError("input layers over more than one variable per fold")
            self._scope_dev = torch.from_numpy(np.ascontiguousarray(self.scope_idx[:, 0])).to(device)
        return self._scope_dev

    def launch_input(self, xt: torch.Tensor, D: int, out: torch.Tensor, B: int, stream: int) -> None:
        raise NotImplementedError

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (F, B, 1) slice of the batch as the reference passes it (circuits.py:66)."""
        if x.dim() != 3 or x.shape[0] != self.num_folds or x.shape[2] != 1:
            raise ValueError(f"expected input of shape (F={self.num_folds}, B, 1), found {tuple(x.shape)}")
        F, B, _ = x.shape
        dev = x.device
        with torch.cuda.device(dev):  # the launches below go to a stream of the input's device
            stream = _stream(dev)
            # the (F, B) slice already is the (D=F, B) staging layout with scope = identity
            if self.wants_float_input:
                xt = x.reshape(F, B).to(torch.float32).contiguous()
            else:
                xt = x.reshape(F, B).to(torch.int32).contiguous()
            saved, self._scope_dev = self._scope_dev, torch.arange(F, dtype=torch.int64, device=dev)
            try:
                self.prepare(stream)
                out = torch.empty((F, B, self.num_output_units), dtype=self.act_dtype, device=dev)
                self.launch_input(xt, F, out, B, stream)
            finally:
                self._scope_dev = saved
        return out

    wants_float_input = False
    can_integrate = False  # whether a marginalised variable (negative category / NaN) is understood


class HipCategoricalLayer(HipInputLayer):
    """``TorchCategoricalLayer`` (layers/input.py:308-434); forward = log_unnormalized_likelihood
    (:399-412) mapped from lse-sum (:276-278)."""

    def __init__(
        self,
        scope_idx,
        num_output_units: int,
        *,
        num_categories: int = 2,
        probs: HipParameter | None = None,
        logits: HipParameter | None = None,
        semiring: str | None = None,
    ) -> None:
        if num_categories <= 0:
            raise ValueError("The number of categories for Categorical distribution must be positive")
        super().__init__(scope_idx, num_output_units, semiring=semiring)
        if self.num_variables != 1:
            raise ValueError("The Categorical layer encodes a univariate distribution")
        self.num_categories = num_categories
        if not ((logits is None) ^ (probs is None)):
            raise ValueError("Exactly one between 'logits' and 'probs' must be specified")
        p = probs if probs is not None else logits
        self._check_param("probs" if probs is not None else "logits", p, (num_output_units, num_categories))
        self.probs, self.logits = probs, logits
        self._table: torch.Tensor | None = None

    can_integrate = True

    @property
    def config(self) -> Mapping[str, Any]:
        return {"num_output_units": self.num_output_units, "num_categories": self.num_categories}

    @property
    def params(self) -> Mapping[str, HipParameter]:
        return {"probs": self.probs} if self.logits is None else {"logits": self.logits}

    def register_batched(self, batch) -> bool:
        src = None if self.probs is None else self.probs.softmax_source()
        if src is None:
            return False
        F, K, C = src.shape
        if (K * (C + 1) + 2 * K) * 4 > 160 * 1024:  # the table job keeps one fold's (K, C) block in LDS
            return False  # -> per-node kernels in `prepare`
        self._table = torch.empty((F, C + 1, K), dtype=torch.float32, device=src.device)  # row C: integral row
        batch.add_log_table(src, self._table)
        self._batched = True
        return True

    def prepare(self, stream: int, batched: bool = False) -> None:
        if batched and self._batched:
            return
        # table (F, C+1, K) = transpose(log(probs())) | transpose(logits())  -- input.py:405-408;
        # row C = the layer's integral (log_partition_function, input.py:414-421)
        p = self.probs if self.probs is not None else self.logits
        v = p.evaluate(stream)
        if v.is_complex():
            raise NotImplementedError("complex categorical parameters")
        F, K, C = v.shape
        if self._table is None or self._table.device != v.device:
            self._table = torch.empty((F, C + 1, K), dtype=torch.float32, device=v.device)
        capi.call(
            "ck_param_transpose_last2", _ptr(v), _ptr(self._table), F, K, C,
            1 if self.probs is not None else 0, C + 1, stream,
        )
        capi.call("ck_param_table_integral_row", _ptr(self._table), F, C, K, 0 if self.probs is not None else 1, stream)

    def launch_input(self, xt, D, out, B, stream) -> None:
        # under complex-lse-sum the real log-likelihood is mapped into the complex semiring (input.py:276-278)
        capi.call(
            "ck_categorical_clog_fwd" if self.is_complex else "ck_categorical_fwd", _ptr(self._table), _ptr(xt), _ptr(self._scope(xt.device)), _ptr(out),
            self.num_folds, B, self.num_output_units, self.num_categories, D, stream,
        )


class HipBinomialLayer(HipCategoricalLayer):
    """``TorchBinomialLayer`` (layers/input.py:437-549): every unit a Binomial(total_count, p_k) over the values
    0 .. total_count.  Evaluated like a Categorical layer with total_count + 1 categories whose log-table is the
    Binomial log-pmf (`ck_param_binomial_table`, the arithmetic of torch.distributions.Binomial.log_prob)."""

    def __init__(self, scope_idx, num_output_units: int, *, total_count: int = 1, probs: HipParameter | None = None,
                 logits: HipParameter | None = None, semiring: str | None = None) -> None:
        if total_count < 0:
            raise ValueError("The number of trials should be non-negative")
        HipInputLayer.__init__(self, scope_idx, num_output_units, semiring=semiring)
        if self.num_variables != 1:
            raise ValueError("The Binomial layer encodes a univariate distribution")
        if not ((logits is None) ^ (probs is None)):
            raise ValueError("Exactly one between 'logits' and 'probs' must be specified")
        self.total_count = total_count
        self.num_categories = total_count + 1
        p = probs if probs is not None else logits
        self._check_param("probs" if probs is not None else "logits", p, (num_output_units,))
        self.probs, self.logits = probs, logits
        self._table = None

    @property
    def config(self) -> Mapping[str, Any]:
        return {"num_output_units": self.num_output_units, "total_count": self.total_count}

    def register_batched(self, batch) -> bool:
        return False

    def prepare(self, stream: int, batched: bool = False) -> None:
        p = self.probs if self.probs is not None else self.logits
        v = p.evaluate(stream)
        if v.is_complex():
            raise NotImplementedError("complex binomial parameters")
        F, K = v.shape
        if self._table is None or self._table.device != v.device:
            self._table = torch.empty((F, self.total_count + 2, K), dtype=torch.float32, device=v.device)
        capi.call("ck_param_binomial_table", _ptr(v.contiguous()), 0 if self.probs is not None else 1, _ptr(self._table),
                  F, K, self.total_count, stream)


class HipEmbeddingLayer(HipInputLayer):
    """``TorchEmbeddingLayer`` (layers/input.py:186-266): ``weight[f, :, x]`` mapped from the
    sum-product semiring, i.e. log (lse-sum) or complex log (complex-lse-sum)."""

    def __init__(
        self,
        scope_idx,
        num_output_units: int,
        *,
        num_states: int = 2,
        weight: HipParameter,
        semiring: str | None = None,
    ) -> None:
        if num_states <= 1:
            raise ValueError("The number of states for Embedding must be at least 2")
        super().__init__(scope_idx, num_output_units, semiring=semiring)
        if self.num_variables != 1:
            raise ValueError("The Embedding layer is defined over exactly one variable")
        self.num_states = num_states
        self._check_param("weight", weight, (num_output_units, num_states))
        self.weight = weight
        self._table: torch.Tensor | None = None

    @property
    def config(self) -> Mapping[str, Any]:
        return {"num_output_units": self.num_output_units, "num_states": self.num_states}

    @property
    def params(self) -> Mapping[str, HipParameter]:
        return {"weight": self.weight}

    def prepare(self, stream: int, batched: bool = False) -> None:
        v = self.weight.evaluate(stream)
        F, K, C = v.shape
        if v.is_complex():
            # complex weights (rules/parameters.py:75-86 compiles DataType.COMPLEX tensors): a complex table, the complex
            # logarithm of its rows (`ck_embedding_clog_c_fwd`) -- under complex-lse-sum only, as in the reference, where the
            # real lse-sum semiring cannot hold a complex value (semiring.py:383-408 works on real tensors)
            if not self.is_complex:
                raise ValueError("complex embedding weights under the real lse-sum semiring")
            if self._table is None or self._table.device != v.device or self._table.dtype != torch.complex64:
                self._table = torch.empty((F, C + 1, K), dtype=torch.complex64, device=v.device)
            capi.call("ck_param_transpose_last2_c", _ptr(v), _ptr(self._table), F, K, C, C + 1, stream)
            capi.call("ck_param_table_integral_row", _ptr(self._table), F, C, 2 * K, 3, stream)
            return
        if self._table is None or self._table.device != v.device or self._table.dtype != torch.float32:
            self._table = torch.empty((F, C + 1, K), dtype=torch.float32, device=v.device)
        capi.call("ck_param_transpose_last2", _ptr(v), _ptr(self._table), F, K, C, 0, C + 1, stream)
        capi.call("ck_param_table_integral_row", _ptr(self._table), F, C, K, 2, stream)

    def launch_input(self, xt, D, out, B, stream) -> None:
        name = "ck_embedding_log_fwd"
        if self.is_complex:
            name = "ck_embedding_clog_c_fwd" if self._table.is_complex() else "ck_embedding_clog_fwd"
        capi.call(name, _ptr(self._table), _ptr(xt), _ptr(self._scope(xt.device)), _ptr(out),
                  self.num_folds, B, self.num_output_units, self.num_states, D, stream)


class HipGaussianLayer(HipInputLayer):
    """``TorchGaussianLayer`` (layers/input.py:564-690), log_unnormalized_likelihood :661-670."""

    wants_float_input = True
    can_integrate = True

    def __init__(
        self,
        scope_idx,
        num_output_units: int,
        *,
        mean: HipParameter,
        stddev: HipParameter,
        log_partition: HipParameter | None = None,
        semiring: str | None = None,
    ) -> None:
        super().__init__(scope_idx, num_output_units, semiring=semiring)
        if self.num_variables != 1:
            raise ValueError("The Gaussian layer encodes a univariate distribution")
        self._check_param("mean", mean, (num_output_units,))
        self._check_param("stddev", stddev, (num_output_units,))
        if log_partition is not None:
            self._check_param("log_partition", log_partition, (num_output_units,))
        self.mean, self.stddev, self.log_partition = mean, stddev, log_partition
        self._vals: tuple | None = None
        self._real_scratch: dict[tuple, torch.Tensor] = {}  # fp32 log-densities under complex-lse-sum (launch_input)

    @property
    def config(self) -> Mapping[str, Any]:
        return {"num_output_units": self.num_output_units}

    @property
    def params(self) -> Mapping[str, HipParameter]:
        p = {"mean": self.mean, "stddev": self.stddev}
        if self.log_partition is not None:
            p["log_partition"] = self.log_partition
        return p

    def prepare(self, stream: int, batched: bool = False) -> None:
        self._vals = (
            self.mean.evaluate(stream),
            self.stddev.evaluate(stream),
            None if self.log_partition is None else self.log_partition.evaluate(stream),
        )

    def launch_input(self, xt, D, out, B, stream) -> None:
        mean, stddev, lz = self._vals
        real = out
        if self.is_complex:  # the real log-density, then its image in the complex semiring (input.py:276-278)
            # The launches below may be RECORDED (ck_program) and replayed later: the scratch block has to outlive this
            # call, so it belongs to the layer (one per output shape / device), never to the caching allocator.
            key = (tuple(out.shape), str(out.device))
            real = self._real_scratch.get(key)
            if real is None:
                if len(self._real_scratch) >= 8:  # (a circuit keeps at most 4 batch sizes bound, oldest evicted first)
                    self._real_scratch.pop(next(iter(self._real_scratch)))
                real = self._real_scratch[key] = torch.empty(out.shape, dtype=torch.float32, device=out.device)
        capi.call(
            "ck_gaussian_fwd", _ptr(mean), _ptr(stddev), _ptr(lz), _ptr(xt), _ptr(self._scope(xt.device)),
            _ptr(real), self.num_folds, B, self.num_output_units, D, stream,
        )
        if self.is_complex:
            capi.call("ck_lse_to_clse", _ptr(real), _ptr(out), real.numel(), stream)


class HipConstantValueLayer(HipLayer):
    """``TorchConstantValueLayer`` (layers/input.py:693-743): ``forward(batch_size)``."""

    def __init__(
        self,
        num_output_units: int,
        *,
        log_space: bool = False,
        value: HipParameter,
        semiring: str | None = None,
        num_folds: int | None = None,
    ) -> None:
        super().__init__(0, num_output_units, semiring=semiring, num_folds=value.num_folds)
        self._check_param("value", value, (num_output_units,))
        self.value = value
        self.log_space = bool(log_space)
        self.scope_idx = np.zeros((self.num_folds, 0), dtype=np.int64)
        self._val: torch.Tensor | None = None

    num_variables = 0

    @property
    def config(self) -> Mapping[str, Any]:
        return {"num_output_units": self.num_output_units, "log_space": self.log_space}

    @property
    def params(self) -> Mapping[str, HipParameter]:
        return {"value": self.value}

    def prepare(self, stream: int, batched: bool = False) -> None:
        self._val = self.value.evaluate(stream)

    def launch_const(self, out: torch.Tensor, B: int, stream: int) -> None:
        v = self._val
        if v.is_complex() and not self.is_complex:
            raise ValueError("complex constant value under the real lse-sum semiring")
        capi.call(
            "ck_constant_fwd", _ptr(v), _ptr(out), self.num_folds, B, self.num_output_units,
            1 if self.log_space else 0, 1 if v.is_complex() else 0, 1 if self.is_complex else 0, stream,
        )

    def forward(self, batch_size: int) -> torch.Tensor:
        dev = self.value.store.device
        with torch.cuda.device(dev):
            stream = _stream(dev)
            self.prepare(stream)
            out = torch.empty((self.num_folds, batch_size, self.num_output_units), dtype=self.act_dtype, device=dev)
            self.launch_const(out, batch_size, stream)
        return out


# =============================================================================================
# inner layers  (layers/inner.py, layers/optimized.py)
# =============================================================================================
class HipInnerLayer(HipLayer):
    """``TorchInnerLayer`` (layers/inner.py:14-64): forward(x: (F, H, B, Ki)) -> (F, B, Ko)."""

    def launch(self, arena: torch.Tensor, row_off: torch.Tensor, out: torch.Tensor, B: int, stream: int) -> None:
        raise NotImplementedError

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[0] != self.num_folds or x.shape[1] != self.arity or x.shape[3] != self.num_input_units:
            raise ValueError(
                f"expected input of shape (F={self.num_folds}, H={self.arity}, B, Ki={self.num_input_units}), "
                f"found {tuple(x.shape)}"
            )
        if x.dtype != self.act_dtype:
            raise ValueError(f"expected {self.act_dtype} activations, found {x.dtype}")
        x = x.contiguous()
        F, H, B, Ki = x.shape
        dev = x.device
        with torch.cuda.device(dev):  # the launches below go to a stream of the input's device
            stream = _stream(dev)
            row_off = (torch.arange(F * H, dtype=torch.int64, device=dev) * (B * Ki)).reshape(F, H)
            self.prepare(stream)
            out = torch.empty((F, B, self.num_output_units), dtype=self.act_dtype, device=dev)
            self.launch(x, row_off, out, B, stream)
        return out


class HipHadamardLayer(HipInnerLayer):
    """``TorchHadamardLayer`` (layers/inner.py:67-135), forward :126-127."""

    def __init__(self, num_input_units: int, arity: int = 2, *, semiring: str | None = None, num_folds: int = 1):
        if arity < 2:
            raise ValueError("The arity should be at least 2")
        super().__init__(num_input_units, num_input_units, arity=arity, semiring=semiring, num_folds=num_folds)

    @property
    def config(self) -> Mapping[str, Any]:
        return {"num_input_units": self.num_input_units, "arity": self.arity}

    def launch(self, arena, row_off, out, B, stream) -> None:
        capi.call(
            "ck_hadamard_fwd", _ptr(arena), _ptr(row_off), _ptr(out), self.num_folds, self.arity, B,
            self.num_input_units, self.esize, stream,
        )


class HipKroneckerLayer(HipInnerLayer):
    """``TorchKroneckerLayer`` (layers/inner.py:138-199), forward :178-187 (any arity >= 2)."""

    def __init__(self, num_input_units: int, arity: int = 2, *, semiring: str | None = None, num_folds: int = 1):
        if arity < 2:
            raise ValueError("The arity should be at least 2")
        super().__init__(num_input_units, num_input_units**arity, arity=arity, semiring=semiring, num_folds=num_folds)

    @property
    def config(self) -> Mapping[str, Any]:
        return {"num_input_units": self.num_input_units, "arity": self.arity}

    def launch(self, arena, row_off, out, B, stream) -> None:
        capi.call(
            "ck_kronecker_fwd", _ptr(arena), _ptr(row_off), _ptr(out), self.num_folds, self.arity, B,
            self.num_input_units, self.esize, stream,
        )


class HipSumLayer(HipInnerLayer):
    """``TorchSumLayer`` (layers/inner.py:202-273): dense (arity 1), general arity, and mixing layers
    (weight graph ending in TorchMixingWeightParameter, nodes.py:847-862)."""

    _mode = capi.CK_SUM_CAT

    # set by HipCircuit(contraction="bf16x3" / "bf16x6"): the launches that have a bf16-piece variant use it (`ck_sum_lse_fwd_v`)
    _contraction = 0

    def __init__(
        self,
        num_input_units: int,
        num_output_units: int,
        arity: int = 1,
        *,
        weight: HipParameter,
        semiring: str | None = None,
        num_folds: int = 1,
    ) -> None:
        if arity < 1:
            raise ValueError("The arity must be a positive integer")
        super().__init__(num_input_units, num_output_units, arity=arity, semiring=semiring, num_folds=num_folds)
        self._check_param("weight", weight, self._weight_shape)
        self.weight = weight
        self._w: torch.Tensor | None = None
        self._w_layout = capi.CK_W_ROWMAJOR  # set by HipCircuit before register_batched
        self._mixing = self._is_mixing()

    @property
    def _weight_shape(self) -> tuple[int, ...]:
        return self.num_output_units, self.num_input_units * self.arity

    @property
    def config(self) -> Mapping[str, Any]:
        return {
            "num_input_units": self.num_input_units,
            "num_output_units": self.num_output_units,
            "arity": self.arity,
        }

    @property
    def params(self) -> Mapping[str, HipParameter]:
        return {"weight": self.weight}

    def _is_mixing(self) -> bool:
        w = self.weight
        if self.is_complex or self._mode != capi.CK_SUM_CAT or not w.tail_is("mixing_weight"):
            return False
        g = w.graph
        fi = g.nodes[-1].inputs[0]
        return fi.ids == [len(g.nodes) - 2] and fi.kind == "none" and self.num_input_units == self.num_output_units

    @property
    def tile32_eligible(self) -> bool:
        """Ki = Ko = 32 product-type real layer whose weight is a plain softmax: may take the
        MFMA-tiled weight layouts written by the batched prologue."""
        return (
            not self._mixing
            and not self.is_complex
            and (self._mode == capi.CK_SUM_PROD or self.arity == 1)
            and self.num_input_units == 32
            and self.num_output_units == 32
            and self.weight.softmax_source() is not None
        )

    def register_batched(self, batch) -> bool:
        if self._mixing and not self.is_complex:
            # the (F, K, H) mixing coefficients: rows of H <= 32 entries in the same batched launch
            src = self.weight.mixing_softmax_source()
            self._w_layout = capi.CK_W_ROWMAJOR
            if src is None or src.shape[-1] > 32:
                return False
            self._w = torch.empty_like(src)
            batch.add_softmax(src, self._w, capi.CK_W_ROWMAJOR)
            self._batched = True
            return True
        src = None if (self._mixing or self.is_complex) else self.weight.softmax_source()
        if src is None:
            self._w_layout = capi.CK_W_ROWMAJOR
            return False
        self._w = torch.empty_like(src)
        batch.add_softmax(src, self._w, self._w_layout)
        self._batched = True
        return True

    def prepare(self, stream: int, batched: bool = False) -> None:
        if batched and self._batched:
            return
        if self._w_layout != capi.CK_W_ROWMAJOR:
            raise capi.HipExtensionError("tiled weight layouts are only produced by the batched prologue")
        if self._mixing:
            self._w = self.weight.evaluate(stream, upto=len(self.weight.graph.nodes) - 2)  # (F, K, H)
        else:
            self._w = self.weight.evaluate(stream)

    def launch(self, arena, row_off, out, B, stream) -> None:
        w = self._w
        if self._mixing:
            capi.call(
                "ck_mixing_lse_fwd", _ptr(arena), _ptr(row_off), _ptr(w), _ptr(out), self.num_folds,
                self.arity, B, self.num_output_units, stream,
            )
        elif self.is_complex:
            capi.call(
                "ck_sum_lse_fwd_c", _ptr(arena), _ptr(row_off), _ptr(w), _ptr(out), self.num_folds, self.arity,
                B, self.num_input_units, self.num_output_units, self._mode, 1 if w.is_complex() else 0, stream,
            )
        else:
            if w.is_complex():
                raise ValueError("complex weights under the real lse-sum semiring")
            capi.call(
                "ck_sum_lse_fwd_v", _ptr(arena), _ptr(row_off), _ptr(w), _ptr(out), self.num_folds, self.arity,
                B, self.num_input_units, self.num_output_units, self._mode, self._w_layout, self._contraction, stream,
            )


class HipCPTLayer(HipSumLayer):
    """``TorchCPTLayer`` (layers/optimized.py:106-178): Hadamard product then dense sum."""

    _mode = capi.CK_SUM_PROD

    @property
    def _weight_shape(self) -> tuple[int, ...]:
        return self.num_output_units, self.num_input_units


class HipTuckerLayer(HipSumLayer):
    """``TorchTuckerLayer`` (layers/optimized.py:17-103): weight (Ko, Ki**arity); one maximum per
    input, the Kronecker product of the shifted inputs is formed on chip."""

    _mode = capi.CK_SUM_KRON

    def __init__(self, num_input_units: int, num_output_units: int, arity: int = 2, *, weight: HipParameter,
                 semiring: str | None = None, num_folds: int = 1) -> None:
        if arity < 2:
            raise ValueError("The arity should be at least 2")
        super().__init__(num_input_units, num_output_units, arity, weight=weight, semiring=semiring, num_folds=num_folds)

    @property
    def _weight_shape(self) -> tuple[int, ...]:
        return self.num_output_units, self.num_input_units**self.arity

    def _is_mixing(self) -> bool:
        return False

    @property
    def tile32_eligible(self) -> bool:
        return False

    # set by HipCircuit(fused_weight_softmax=True): a softmax(theta) weight is not evaluated by the prologue at all -- the
    # stream-K Tucker launch reads the logits and normalises them online (`ck_tucker_logits_fwd`: running row maximum
    # and sum beside the accumulators), so the (F, Ko, Ki^2) weights are read once per forward and never written.  Not
    # for training: the backward kernels read `_w`.
    _logits_ok = False
    _theta: torch.Tensor | None = None
    _use_logits = False
    # set by HipCircuit(contraction="bf16x3" / "bf16x6"): the stream-K launch contracts on the bf16 matrix pipe (a labelled
    # variant, `ck_tucker_fwd`); where that launch does not apply (a large batch) the layer runs in exact fp32
    _contraction = 0

    def register_batched(self, batch) -> bool:
        self._theta = None
        src = None if self.is_complex else self.weight.softmax_source()
        n = self.num_input_units ** self.arity
        if (self._logits_ok and src is not None and self.arity == 2 and self.num_input_units in (32, 64)
                and src.shape[-1] == n and src.is_contiguous() and src.data_ptr() % 16 == 0):
            self._theta = src
            self._w_layout = capi.CK_W_ROWMAJOR
            self._batched = True  # (nothing for the prologue to do)
            return True
        return super().register_batched(batch)

    def prepare(self, stream: int, batched: bool = False) -> None:
        self._use_logits = bool(batched and self._batched and self._theta is not None)
        if not self._use_logits:
            if batched and self._batched:
                return
            self._w = self.weight.evaluate(stream)  # (the per-node path evaluates the normalised weights)

    def launch(self, arena, row_off, out, B, stream) -> None:
        ct = self._contraction if (not self.is_complex and self.arity == 2 and self.num_input_units in (32, 64)) else 0
        args = (self.num_folds, B, self.num_input_units, self.num_output_units)
        if self._use_logits:
            # the stream-K launch reads the logits and normalises them online.  With many tiles per resident workgroup (a large
            # batch) the exact launch refuses them (NotImplementedError: one workgroup per tile would exponentiate the weights
            # once per 128 rows) and the normalised weights are written after all; the bf16 variants take the logits at any
            # size (measured: as fast as on written weights, and the prologue's pass over them is saved)
            try:
                capi.call("ck_tucker_fwd", _ptr(arena), _ptr(row_off), _ptr(self._theta), _ptr(out), *args, 1, ct, stream)
                return
            except NotImplementedError:
                pass
            self._w = self.weight.evaluate(stream)
        if ct and self._w_layout == capi.CK_W_ROWMAJOR:  # the labelled variants of the stream-K launch (any batch size)
            try:
                capi.call("ck_tucker_fwd", _ptr(arena), _ptr(row_off), _ptr(self._w), _ptr(out), *args, 0, ct, stream)
                return
            except NotImplementedError:
                pass
        super().launch(arena, row_off, out, B, stream)


class HipTensorDotLayer(HipInnerLayer):
    """``TorchTensorDotLayer`` (layers/optimized.py:181-300)."""

    def __init__(
        self,
        num_input_units: int,
        num_output_units: int,
        *,
        weight: HipParameter,
        semiring: str | None = None,
        num_folds: int = 1,
    ) -> None:
        super().__init__(num_input_units, num_output_units, arity=1, semiring=semiring, num_folds=num_folds)
        ws = tuple(weight.shape)
        ok = (
            weight.num_folds == num_folds
            and len(ws) == 2
            and num_input_units % ws[1] == 0
            and num_output_units == ws[0] * (num_input_units // ws[1])
        )
        if not ok:
            raise ValueError(
                f"Expected number of folds {num_folds} and shape (K_k, K_j) for 'weight', where "
                f"{num_input_units} = K_jK_q and {num_output_units} = K_qK_k, but found "
                f"{weight.num_folds} and {ws}, respectively"
            )
        self.weight = weight
        self._num_contract_units = ws[1]
        self._num_batch_units = num_input_units // ws[1]
        self._w: torch.Tensor | None = None

    @property
    def config(self) -> Mapping[str, Any]:
        return {"num_input_units": self.num_input_units, "num_output_units": self.num_output_units}

    @property
    def params(self) -> Mapping[str, HipParameter]:
        return {"weight": self.weight}

    def prepare(self, stream: int, batched: bool = False) -> None:
        self._w = self.weight.evaluate(stream)

    def launch(self, arena, row_off, out, B, stream) -> None:
        w = self._w
        Kk = int(self.weight.shape[0])
        if self.is_complex:
            capi.call(
                "ck_tensordot_lse_fwd_c", _ptr(arena), _ptr(row_off), _ptr(w), _ptr(out), self.num_folds, B,
                self._num_contract_units, self._num_batch_units, Kk, 1 if w.is_complex() else 0, stream,
            )
        else:
            if w.is_complex():
                raise ValueError("complex weights under the real lse-sum semiring")
            capi.call(
                "ck_tensordot_lse_fwd", _ptr(arena), _ptr(row_off), _ptr(w), _ptr(out), self.num_folds, B,
                self._num_contract_units, self._num_batch_units, Kk, stream,
            )


LAYER_CLASSES: dict[str, type] = {
    "categorical": HipCategoricalLayer,
    "binomial": HipBinomialLayer,
    "gaussian": HipGaussianLayer,
    "embedding": HipEmbeddingLayer,
    "constant": HipConstantValueLayer,
    "sum": HipSumLayer,
    "cpt": HipCPTLayer,
    "hadamard": HipHadamardLayer,
    "kronecker": HipKroneckerLayer,
    "tensordot": HipTensorDotLayer,
    "tucker": HipTuckerLayer,
}


def layer_from_spec(spec, store, semiring: str) -> HipLayer:
    """Instantiate the HIP layer of a `plan.LayerSpec` -- same call shape the reference's folding
    uses: ``cls(semiring=..., **config, num_folds=..., **params)`` (compiler.py:398-406)."""
    if spec.type not in LAYER_CLASSES:
        raise NotImplementedError(f"layer type {spec.type!r} is not on the HIP hot path (SURVEY.md section 8)")
    cls = LAYER_CLASSES[spec.type]
    params = {pn: HipParameter(pg, store) for pn, pg in spec.params.items()}
    cfg = dict(spec.config)
    if issubclass(cls, HipInputLayer):
        return cls(spec.scope_idx, semiring=semiring, **cfg, **params)
    if cls is HipConstantValueLayer:
        return cls(semiring=semiring, **cfg, **params)
    return cls(semiring=semiring, num_folds=spec.num_folds, **cfg, **params)
