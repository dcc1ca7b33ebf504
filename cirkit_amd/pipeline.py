"""``compile()`` entry of the HIP backend -- the counterpart of ``cirkit.pipeline.compile``
(cirkit/pipeline.py:298-301) and ``PipelineContext`` (:22-65).

cirkit selects its backend with a hard-coded ``if backend == "torch"`` (pipeline.py:348-356), so
the drop-in point is *behind* the symbolic compiler: the reference lowers a symbolic circuit to a
folded ``TorchCircuit`` (symbolic -> layers -> optimise -> fold; host-only, once per circuit), and
this module turns that into a `HipCircuit` that owns the per-batch forward.  Accepted inputs:

* a compiled reference ``TorchCircuit`` (any object with its ``address_book`` surface);
* a symbolic ``cirkit.symbolic.circuit.Circuit`` -- compiled with the reference compiler first
  (needs ``cirkit`` importable);
* a `Plan` (+ tensors), e.g. loaded from a fixture or built by `cirkit_amd.templates`.
"""

from __future__ import annotations

from typing import Any, Mapping

from .circuit import HipCircuit
from .plan import Plan, plan_from_torch_circuit, tensor_table


class HipPipelineContext:
    """Mirror of ``PipelineContext(backend, semiring, fold, optimize)`` for ``backend="hip"``."""

    def __init__(
        self,
        *,
        semiring: str = "lse-sum",
        fold: bool = True,
        optimize: bool = True,
        device: str = "cuda:0",
        use_graph: bool = True,
    ) -> None:
        if semiring not in ("lse-sum", "complex-lse-sum"):
            raise ValueError(f"semiring {semiring!r} is not evaluated by the HIP backend")
        self.semiring, self.fold, self.optimize = semiring, fold, optimize
        self.device, self.use_graph = device, use_graph
        self._table = tensor_table()  # shared so products / conjugates point at the same weights
        self._stores: dict[str, Any] = {}

    def compile(self, circuit: Any, tensors: Mapping[str, Any] | None = None) -> HipCircuit:
        if isinstance(circuit, Plan):
            if tensors is None:
                raise ValueError("compiling a Plan needs its parameter tensors")
            return HipCircuit(circuit, tensors, device=self.device, use_graph=self.use_graph)
        if not hasattr(circuit, "address_book"):
            # a symbolic circuit: lower it with the reference's own compiler
            try:
                from cirkit.pipeline import PipelineContext  # type: ignore
            except ImportError as e:  # pragma: no cover - depends on the environment
                raise ImportError(
                    "compiling a symbolic circuit needs april-tools/cirkit importable; "
                    "pass a compiled TorchCircuit or a Plan instead"
                ) from e
            ctx = PipelineContext(
                backend="torch", semiring=self.semiring, fold=self.fold, optimize=self.optimize
            )
            circuit = ctx.compile(circuit)
        plan, tvals = plan_from_torch_circuit(circuit, table=self._table)
        if plan.semiring != self.semiring:
            raise ValueError(f"circuit was compiled under {plan.semiring!r}, context is {self.semiring!r}")
        # (parameters extracted from a reference circuit stay SHARED with it: no padding copies, and -- since writes through
        # `.data` or foreign kernels are invisible to the store -- derived parameters re-evaluated at the start of every forward)
        return HipCircuit(plan, tvals, device=self.device, use_graph=self.use_graph, pad_units=False, params_at_end=False)


def compile_unfolded(plan: Plan, tensors: Mapping[str, Any], **ctx_kwargs: Any) -> HipCircuit:
    """An UNFOLDED plan (one fold per layer, e.g. extracted with ``fold=False, optimize=False`` or
    written by hand) -> optimise + fold natively (cirkit_amd/compiler.py) -> `HipCircuit`."""
    from .compiler import compile_plan

    ctx = HipPipelineContext(**ctx_kwargs)
    plan, vals = compile_plan(plan, tensors, fold=ctx.fold, optimize=ctx.optimize)
    return ctx.compile(plan, vals)


def compile(circuit: Any, tensors: Mapping[str, Any] | None = None, **ctx_kwargs: Any) -> HipCircuit:
    """``cirkit_amd.pipeline.compile(sc)`` -- see the module docstring."""
    return HipPipelineContext(**ctx_kwargs).compile(circuit, tensors)
