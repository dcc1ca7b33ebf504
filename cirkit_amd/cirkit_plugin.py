"""The HIP layers as a cirkit plugin: layer compilation rules that return ``TorchLayer``
SUBCLASSES whose ``forward`` calls the C ABI, registered on a ``PipelineContext``::

    import cirkit_amd.cirkit_plugin as plugin
    ctx = plugin.HipLayersContext(semiring="lse-sum", fold=True, optimize=True)   # or plugin.register(existing_ctx, its_compiler)
    cc = ctx.compile(symbolic_circuit).to("cuda")      # every layer is a Hip* subclass of the reference's class
    cc(x)                                              # the reference's interpreter loop and gather, HIP layer kernels

This module imports cirkit (it is the half that lives in a process where the reference is installed); everything it
calls -- `cirkit_amd.layer_ops` -- does not.  What it relies on in the reference:

* layer compilation rules are looked up by the symbolic layer class annotated on their last parameter
  (cirkit/backend/compiler.py:101-113) and registered with ``PipelineContext.add_layer_compilation_rule``
  (cirkit/pipeline.py:110-116); each rule here runs the reference's own rule and re-instantiates its result as the
  subclass (same ``config``, same parameter graphs);
* the optimiser matches patterns with ``isinstance`` (torch/compiler.py:733-736), so the subclasses are fused like
  their bases -- but the apply functions build stock ``TorchCPTLayer`` / ``TorchTuckerLayer`` / ``TorchSumLayer`` /
  ``TorchTensorDotLayer`` objects, so the "fuse" and "shatter" registries (torch/compiler.py:226-227) get wrapped apply
  functions that convert what the stock ones return;
* folding re-instantiates ``type(layers[0])(semiring=..., **config, **folded_params)`` (torch/compiler.py:374-406): the
  subclasses keep the constructors of their bases, so they survive it.

Parameters stay the reference's ``TorchParameter`` graphs (evaluated by torch on the same device); the layer forward --
the gather from the category table, the log-einsum-exp, the products -- is the HIP kernel.  Semirings: lse-sum and
complex-lse-sum; anything else raises (no fallback to the stock forward).

**Training.**  Under the real lse-sum semiring the Sum / CP-T / Tucker / Hadamard / Kronecker / Categorical / Gaussian
forwards are ``torch.autograd.Function``s (`cirkit_amd.layer_ops`): ``loss = -cc(x).mean(); loss.backward(); opt.step()``
-- the reference's training loop, notebooks/learning-a-circuit.ipynb -- runs unchanged; autograd differentiates the
reference's parameter graphs and the gather between layers, the hand-written kernels of ck_backward.hip supply each
layer's d/dx and d/dW.  Under complex-lse-sum the sum layers (`ck_sum_lse_bwd_c`), Hadamard, Embedding and ConstantValue layers do the
same (squared circuits train through the reference's loop); only complex Kronecker layers record no graph: they RAISE
when gradients are enabled and an input or parameter requires them (`layer_ops._forward_only`) -- evaluate those under
``torch.no_grad()``.  The fast path for training a whole plan is `cirkit_amd.training.HipTrainer` (b4 level).
"""

from __future__ import annotations

import weakref
from typing import Any

import torch
from cirkit.backend.torch.layers import TorchHadamardLayer, TorchKroneckerLayer, TorchLayer, TorchSumLayer, TorchTuckerLayer
from cirkit.backend.torch.layers.input import (
    TorchCategoricalLayer,
    TorchConstantLayer,
    TorchConstantValueLayer,
    TorchEmbeddingLayer,
    TorchGaussianLayer,
    TorchInputLayer,
)
from cirkit.backend.torch.layers.optimized import TorchCPTLayer, TorchTensorDotLayer
from cirkit.backend.torch.optimization.registry import LayerOptMatch
from cirkit.pipeline import PipelineContext
from cirkit.backend.torch.rules.layers import DEFAULT_LAYER_COMPILATION_RULES
from cirkit.backend.torch.semiring import ComplexLSESumSemiring, LSESumSemiring

from . import _capi as capi
from . import layer_ops as ops

__all__ = ["register", "HipLayersContext", "to_hip_layer", "HIP_LAYER_CLASSES"]


def _complex(layer: TorchLayer) -> bool:
    if layer.semiring is ComplexLSESumSemiring:
        return True
    if layer.semiring is LSESumSemiring:
        return False
    raise NotImplementedError(
        f"semiring {getattr(layer.semiring, '__name__', layer.semiring)!r} is not evaluated by the HIP layers "
        "(supported: lse-sum, complex-lse-sum)")


class HipSumLayer(TorchSumLayer):
    """``TorchSumLayer`` whose forward is `ck_sum_lse_fwd` in CK_SUM_CAT mode (inner.py:266-273 + semiring.py:383-408)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _complex(self)
        return ops.sum_lse(x, self.weight(), capi.CK_SUM_CAT)


class HipCPTLayer(TorchCPTLayer):
    """``TorchCPTLayer`` (optimized.py:171-178) on `ck_sum_lse_fwd` in CK_SUM_PROD mode."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _complex(self)
        return ops.sum_lse(x, self.weight(), capi.CK_SUM_PROD)


class HipTuckerLayer(TorchTuckerLayer):
    """``TorchTuckerLayer`` (optimized.py:89-103) on `ck_sum_lse_fwd` in CK_SUM_KRON mode."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _complex(self)
        return ops.sum_lse(x, self.weight(), capi.CK_SUM_KRON)


class HipTensorDotLayer(TorchTensorDotLayer):
    """``TorchTensorDotLayer`` (optimized.py:287-300) on `ck_tensordot_lse_fwd`."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _complex(self)
        return ops.tensordot_lse(x, self.weight(), self._num_contract_units, self._num_batch_units)


class HipHadamardLayer(TorchHadamardLayer):
    """``TorchHadamardLayer`` (inner.py:126-127) on `ck_hadamard_fwd`."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _complex(self)
        return ops.hadamard(x)


class HipKroneckerLayer(TorchKroneckerLayer):
    """``TorchKroneckerLayer`` (inner.py:178-187) on `ck_kronecker_fwd`."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _complex(self)
        return ops.kronecker(x)


class HipCategoricalLayer(TorchCategoricalLayer):
    """``TorchCategoricalLayer`` whose log-likelihood gather (input.py:399-412) is `ck_categorical_fwd`."""

    def log_unnormalized_likelihood(self, x: torch.Tensor) -> torch.Tensor:
        logits = torch.log(self.probs()) if self.logits is None else self.logits()
        return ops.categorical_log_likelihood(x, logits)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        cplx = _complex(self)
        y = self.log_unnormalized_likelihood(x)
        return y.to(torch.complex64) if cplx else y


class HipGaussianLayer(TorchGaussianLayer):
    """``TorchGaussianLayer`` whose log-density (input.py:661-670) is `ck_gaussian_fwd`."""

    def log_unnormalized_likelihood(self, x: torch.Tensor) -> torch.Tensor:
        return ops.gaussian_log_likelihood(x, self.mean(), self.stddev(),
                                           None if self.log_partition is None else self.log_partition())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        cplx = _complex(self)
        y = self.log_unnormalized_likelihood(x)
        return y.to(torch.complex64) if cplx else y


class HipEmbeddingLayer(TorchEmbeddingLayer):
    """``TorchEmbeddingLayer`` (input.py:258-266) on `ck_embedding_log_fwd` / `ck_embedding_clog_fwd`."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.embedding(x, self.weight(), complex_out=_complex(self))


class HipConstantValueLayer(TorchConstantValueLayer):
    """``TorchConstantValueLayer`` (input.py:739-743) on `ck_constant_fwd`."""

    def forward(self, batch_size: int) -> torch.Tensor:
        return ops.constant_value(self.value(), batch_size, log_space=self.log_space, complex_out=_complex(self))


# reference class -> the subclass that replaces it (exact type match: a subclass of a subclass is left alone)
HIP_LAYER_CLASSES: dict[type, type] = {
    TorchSumLayer: HipSumLayer,
    TorchCPTLayer: HipCPTLayer,
    TorchTuckerLayer: HipTuckerLayer,
    TorchTensorDotLayer: HipTensorDotLayer,
    TorchHadamardLayer: HipHadamardLayer,
    TorchKroneckerLayer: HipKroneckerLayer,
    TorchCategoricalLayer: HipCategoricalLayer,
    TorchGaussianLayer: HipGaussianLayer,
    TorchEmbeddingLayer: HipEmbeddingLayer,
    TorchConstantValueLayer: HipConstantValueLayer,
}


def to_hip_layer(layer: TorchLayer) -> TorchLayer:
    """Re-instantiate a stock reference layer as its HIP subclass: same ``config``, same parameter graphs, same
    sub-modules -- exactly how folding re-instantiates layers (torch/compiler.py:374-406).  Layers without a HIP subclass
    (and layers that already are one) are returned unchanged."""
    cls = HIP_LAYER_CLASSES.get(type(layer))
    if cls is None:
        return layer
    # the keyword arguments folding uses (torch/compiler.py:378-403): the configuration, the scope indices of an input
    # layer (none for constant layers) or the number of folds of an inner layer, the parameters, the sub-modules
    kwargs: dict[str, Any] = dict(layer.config)
    if isinstance(layer, TorchInputLayer):
        if not isinstance(layer, TorchConstantLayer):
            kwargs["scope_idx"] = layer.scope_idx
    else:
        kwargs["num_folds"] = layer.num_folds
    kwargs.update(layer.params)
    kwargs.update({n: to_hip_layer(m) for n, m in layer.sub_modules.items()})
    return cls(semiring=layer.semiring, **kwargs)


# Compilers whose layers are converted.  The reference hands its module-level DEFAULT_LAYER_COMPILATION_RULES dict to
# every compiler's registry WITHOUT copying it (backend/registry.py:26-27, torch/compiler.py:118), so a rule added through
# the public ``add_layer_compilation_rule`` is seen by every other context of the process: the rules below therefore act
# as the reference's own rule unless the compiler that calls them was enabled here (no private registry state is touched).
_ENABLED: "weakref.WeakSet" = weakref.WeakSet()


def _layer_rule(default_rule):
    """The reference's own compilation rule, followed (for enabled compilers) by the conversion of its result."""
    ann = dict(default_rule.__annotations__)

    def rule(compiler, sl):
        out = default_rule(compiler, sl)
        return to_hip_layer(out) if compiler in _ENABLED else out

    rule.__annotations__ = ann
    rule._cirkit_amd = True
    rule.__name__ = f"hip_{default_rule.__name__}"
    rule.__doc__ = f"{default_rule.__name__} of the reference, returning the HIP subclass of its layer."
    return rule


def _opt_rule(default_apply):
    def apply(compiler, match: LayerOptMatch):
        return tuple(to_hip_layer(l) for l in default_apply(compiler, match))

    apply.__annotations__ = {"compiler": Any, "match": LayerOptMatch, "return": tuple}
    apply._cirkit_amd = True
    apply.__name__ = f"hip_{default_apply.__name__}"
    return apply


def _enable(ctx, compiler) -> None:
    """Public calls only: ``PipelineContext.add_layer_compilation_rule`` (pipeline.py:110-116) for every rule of the
    public DEFAULT_LAYER_COMPILATION_RULES table, ``retrieve_layer_optimization_registry`` (torch/compiler.py:226-227)
    + ``signatures`` / ``retrieve_rule`` / ``add_rule`` (backend/registry.py) for the "fuse" and "shatter" apply functions
    (those registries are per-compiler copies in the reference)."""
    _ENABLED.add(compiler)
    for signature, rule in list(DEFAULT_LAYER_COMPILATION_RULES.items()):
        if not getattr(rule, "_cirkit_amd", False):
            ctx.add_layer_compilation_rule(_layer_rule(rule))
    for kind in ("fuse", "shatter"):
        registry = compiler.retrieve_layer_optimization_registry(kind)
        for pattern in list(registry.signatures):
            apply = registry.retrieve_rule(pattern)
            if not getattr(apply, "_cirkit_amd", False):
                registry.add_rule(_opt_rule(apply), signature=pattern)


class HipLayersContext(PipelineContext):
    """A ``PipelineContext`` (torch backend) whose compiled layers are the HIP subclasses -- the recommended entry point::

        ctx = HipLayersContext(semiring="lse-sum", fold=True, optimize=True)
        cc = ctx.compile(symbolic_circuit).to("cuda")

    A subclass reaches its base's compiler the way the base itself does; nothing else of the reference's state is used."""

    def __init__(self, backend: str = "torch", **backend_kwargs: Any) -> None:
        if backend != "torch":
            raise NotImplementedError("the HIP layers subclass the torch backend's layers")
        super().__init__(backend=backend, **backend_kwargs)
        _enable(self, self._compiler)


def register(ctx, compiler):
    """Enable the HIP layers on an EXISTING ``PipelineContext`` (torch backend) whose compiler the caller holds, and return the
    context: every layer compilation rule of the reference is replaced by one that returns the HIP subclass, and the apply
    functions of the compiler's "fuse" and "shatter" optimisation registries are wrapped the same way.  ``PipelineContext``
    exposes no accessor for its compiler, so this function does not look for it: `HipLayersContext` (a subclass, which
    reaches its base's compiler as the base does) is the entry point that needs nothing from the caller."""
    _enable(ctx, compiler)
    return ctx
