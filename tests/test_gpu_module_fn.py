"""Row b3: `HipModuleFn` driven through an ``evaluate(x, module_fn)`` interpreter loop on the GPU.

The GPU box has no cirkit, so the loop is a stand-in that restates the reference's -- ``TorchDiAcyclicGraph.evaluate``
(graph/modules.py:303-335) over ``LayerAddressBook.lookup`` (circuits.py:30-71): per entry, concatenate the producers'
outputs along the fold axis, index with the fold index, call ``module_fn(module, x)``; input layers receive
``x[..., scope_idx].permute(1, 0, 2)``, constant layers the batch size; the entry without a module returns the output.
The stand-in circuit is built from a committed plan fixture (tests/golden); the result is compared with the golden
output of the REAL reference on the same inputs."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import load_case  # noqa: E402

pytestmark = pytest.mark.gpu


class _Module:
    """A stand-in for a folded TorchLayer: only identity matters to the hook (plus what the loop itself reads)."""

    def __init__(self, spec):
        self.spec = spec
        self.scope_idx = None if spec.scope_idx is None else torch.from_numpy(np.asarray(spec.scope_idx))
        self.num_variables = 0 if spec.scope_idx is None else int(np.asarray(spec.scope_idx).shape[-1])

    def __call__(self, *a):  # the loop must never fall back to the module itself
        raise AssertionError("module_fn was not used")


class _Entry:
    def __init__(self, module, in_module_ids, in_fold_idx):
        self.module, self.in_module_ids, self.in_fold_idx = module, in_module_ids, in_fold_idx


def _torch_index(fi, device):
    """FoldIndex -> what the reference stores in an address-book entry (folding.py:234-241)."""
    if fi.kind == "array":
        return torch.from_numpy(np.asarray(fi.array)).to(device)
    if fi.kind == "unsq0":
        return (None,)
    if fi.kind == "unsq1":
        return (slice(None), None)
    return slice(None)


class StandInCircuit:
    """address_book + evaluate(): the duck-typed surface of a compiled TorchCircuit that the module_fn hook needs."""

    def __init__(self, plan, device):
        self.device = device
        self.modules = [_Module(s) for s in plan.layers]
        self.address_book = []
        for m, s in zip(self.modules, plan.layers):
            if s.inputs is None:
                self.address_book.append(_Entry(m, [], []))
            else:
                self.address_book.append(_Entry(m, [list(s.inputs.ids)], [_torch_index(s.inputs, device)]))
        self.address_book.append(_Entry(None, [list(plan.output.ids)], [_torch_index(plan.output, device)]))

    def lookup(self, module_outputs, in_graph):  # circuits.py:30-71
        for entry in self.address_book:
            layer = entry.module
            if entry.in_module_ids:
                ids, idx = entry.in_module_ids[0], entry.in_fold_idx[0]
                x = module_outputs[ids[0]] if len(ids) == 1 else torch.cat([module_outputs[i] for i in ids], dim=0)
                yield layer, (x[idx],)
                continue
            if layer.num_variables:
                yield layer, (in_graph[..., layer.scope_idx.to(in_graph.device)].permute(1, 0, 2),)
                continue
            yield layer, (1 if in_graph is None else in_graph.shape[0],)

    def evaluate(self, x, module_fn):  # graph/modules.py:303-335
        module_outputs = []
        for module, inputs in self.lookup(module_outputs, x):
            if module is None:
                (output,) = inputs
                return output
            module_outputs.append(module(*inputs) if module_fn is None else module_fn(module, *inputs))
        raise RuntimeError("The address book is malformed")

    def __call__(self, x, module_fn):  # circuits.py:272-278
        y = self.evaluate(x, module_fn).transpose(0, 1)
        return y


@pytest.mark.parametrize("case", ["cfg1_rbt8", "cfg2_qt784", "cfg4_pd784", "cfg5_sos_c_k32"])
def test_module_fn_hook_through_an_interpreter_loop(hip_device, case):
    from cirkit_amd.integration import HipModuleFn

    plan, tensors, g = load_case(case)
    x = torch.from_numpy(g["x"])
    x = (x.to(torch.float32) if case.startswith("cfg4") else x.to(torch.int64)).to(hip_device)
    circ = StandInCircuit(plan, hip_device)
    fn = HipModuleFn.from_plan(plan, tensors, circ.modules, device=hip_device)
    y = circ(x, fn).cpu()
    ref = torch.from_numpy(g["y_c64"] if "y_c64" in g else g["y_f32"])
    assert y.shape == ref.shape
    if torch.is_complex(ref):
        assert float((y.real - ref.real).abs().max()) <= 1e-4 * float(ref.real.abs().max())
    else:
        assert float(((y - ref).abs() / ref.abs().clamp_min(1e-30)).max()) <= 1e-4
    # every module went through the hook; an unknown module is an error, not a fallback
    with pytest.raises(KeyError):
        fn(_Module(plan.layers[0]), x)
