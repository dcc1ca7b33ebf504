"""Drop-in hooks for a process that has april-tools/cirkit importable (SURVEY.md section 8 b).

Nothing here imports ``cirkit``: the reference objects are driven by duck-typing, so the module also
loads where the reference is absent.

b3  `HipModuleFn(circuit)` -- a ``ModuleEvalFunctional`` (cirkit/backend/torch/graph/modules.py:
    224-237): ``circuit.evaluate(x, module_fn=HipModuleFn(circuit))`` keeps the reference's
    interpreter loop and gather (graph/modules.py:303-335, circuits.py:30-71) but evaluates every
    layer with the HIP kernels through the per-layer ``forward`` contract.  Zero patches to cirkit.
b4  `to_hip(circuit)` -- replace the whole forward: extract the folded plan from the compiled
    ``TorchCircuit`` and return a `HipCircuit` (fused gathers, recorded launch list, leaf fusion).
b2  layer compilation rules returning ``TorchLayer`` subclasses that call the C ABI live in
    `cirkit_amd/cirkit_plugin.py` (it imports cirkit; ``plugin.HipLayersContext``, pipeline.py:110-116).
"""

from __future__ import annotations

from typing import Any

import torch

from .circuit import HipCircuit
from .layers import HipConstantValueLayer, HipInputLayer, HipLayer, layer_from_spec
from .parameters import TensorStore
from .plan import plan_from_torch_circuit, tensor_table


def to_hip(circuit: Any, *, device: str | torch.device = "cuda:0", **kw: Any) -> HipCircuit:
    """A compiled reference ``TorchCircuit`` -> `HipCircuit` with the same parameters.  When the
    reference circuit already lives on `device` the parameter storage is shared (in-place optimiser
    steps on the reference's ``nn.Parameter``s are seen by the next HIP forward).  That is why unit counts are NOT
    padded to multiples of 32 here by default (padding copies the parameters into larger tensors): pass
    ``pad_units=True`` to trade the sharing for the MFMA tiles when the widths are not multiples of 32."""
    plan, tensors = plan_from_torch_circuit(circuit)
    kw.setdefault("pad_units", False)
    # The storage is shared with arbitrary torch code: a write through `p.data` or by a foreign kernel bumps no version
    # counter, so whether the derived parameters of the previous forward are still valid cannot be known here -- they are
    # re-evaluated at the START of every forward, exactly as the reference does (parameters/parameter.py:180-188).  The
    # native-plan `HipCircuit` owns its store and defaults to evaluating them at the end of the previous forward.
    kw.setdefault("params_at_end", False)
    return HipCircuit(plan, tensors, device=device, **kw)


class HipModuleFn:
    """``module_fn`` for ``TorchDiAcyclicGraph.evaluate``: ``fn(module, *inputs) -> Tensor``."""

    def __init__(self, circuit: Any, *, device: str | torch.device = "cuda:0") -> None:
        table = tensor_table()
        plan, tensors = plan_from_torch_circuit(circuit, table=table)
        entries = [e for e in circuit.address_book if e.module is not None]
        self._bind(plan, tensors, [e.module for e in entries], device)

    @classmethod
    def from_plan(cls, plan: Any, tensors: Any, modules: list, *, device: str | torch.device = "cuda:0") -> "HipModuleFn":
        """The hook for an interpreter loop whose modules are `modules` (in address-book order) and whose folded plan
        and parameter values are already known -- e.g. extracted earlier, or loaded from a plan file."""
        self = cls.__new__(cls)
        self._bind(plan, tensors, list(modules), device)
        return self

    def _bind(self, plan: Any, tensors: Any, modules: list, device: str | torch.device) -> None:
        self.device = torch.device(device)
        self.plan = plan
        self.store = TensorStore(self.device)
        self.store.update(tensors)
        self._by_module: dict[int, HipLayer] = {}
        if len(modules) != len(plan.layers):
            raise ValueError("address book and plan disagree")
        for m, spec in zip(modules, plan.layers):
            self._by_module[id(m)] = layer_from_spec(spec, self.store, plan.semiring)

    def layer_of(self, module: Any) -> HipLayer:
        try:
            return self._by_module[id(module)]
        except KeyError as e:
            raise KeyError(f"{type(module).__name__} is not a layer of the circuit this hook was built for") from e

    def __call__(self, module: Any, *inputs: Any) -> torch.Tensor:
        layer = self.layer_of(module)
        if isinstance(layer, HipConstantValueLayer):
            (batch_size,) = inputs
            return layer.forward(int(batch_size))
        (x,) = inputs
        if isinstance(layer, HipInputLayer):
            return layer.forward(x.to(self.device))
        return layer.forward(x.to(self.device))
