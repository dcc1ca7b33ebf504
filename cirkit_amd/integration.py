"""Drop-in hooks for a process that has april-tools/cirkit importable (SURVEY.md section 8 b).

Nothing here imports ``cirkit``: the reference objects are driven by duck-typing, so the module also
loads where the reference is absent.

b3  `HipModuleFn(circuit)` -- a ``ModuleEvalFunctional`` (cirkit/backend/torch/graph/modules.py:
    224-237): ``circuit.evaluate(x, module_fn=HipModuleFn(circuit))`` keeps the reference's
    interpreter loop and gather (graph/modules.py:303-335, circuits.py:30-71) but evaluates every
    layer with the HIP kernels through the per-layer ``forward`` contract.  Zero patches to cirkit.
b4  `to_hip(circuit)` -- replace the whole forward: extract the folded plan from the compiled
    ``TorchCircuit`` and return a `HipCircuit` (fused gathers, recorded launch list, leaf fusion).
b2  `layer_rule_for(...)` -- the shape of a layer compilation rule
    (``PipelineContext.add_layer_compilation_rule``, pipeline.py:110-116); see INTEGRATION.md for the
    stub a cirkit maintainer would add.
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
    return HipCircuit(plan, tensors, device=device, **kw)


class HipModuleFn:
    """``module_fn`` for ``TorchDiAcyclicGraph.evaluate``: ``fn(module, *inputs) -> Tensor``."""

    def __init__(self, circuit: Any, *, device: str | torch.device = "cuda:0") -> None:
        self.device = torch.device(device)
        table = tensor_table()
        self.plan, tensors = plan_from_torch_circuit(circuit, table=table)
        self.store = TensorStore(self.device)
        self.store.update(tensors)
        self._by_module: dict[int, HipLayer] = {}
        entries = [e for e in circuit.address_book if e.module is not None]
        if len(entries) != len(self.plan.layers):
            raise ValueError("address book and extracted plan disagree")
        for e, spec in zip(entries, self.plan.layers):
            self._by_module[id(e.module)] = layer_from_spec(spec, self.store, self.plan.semiring)

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


def layer_rule_for(hip_layer_cls: type, symbolic_layer_cls: type):
    """Build a layer compilation rule ``rule(compiler, sl: symbolic_layer_cls)`` whose registry key is
    the annotation of its last parameter (cirkit/backend/compiler.py:101-113).  The rule body a
    maintainer writes is in INTEGRATION.md; this helper only fixes the signature convention."""

    def rule(compiler: Any, sl: Any):  # pragma: no cover - needs cirkit
        raise NotImplementedError(
            "bind this rule inside cirkit (see INTEGRATION.md): it must return a TorchLayer subclass "
            f"that forwards to {hip_layer_cls.__name__}"
        )

    rule.__annotations__ = {"compiler": Any, "sl": symbolic_layer_cls, "return": Any}
    return rule
