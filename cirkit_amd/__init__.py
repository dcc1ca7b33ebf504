"""cirkit_amd -- MI355X-native evaluation backend for cirkit's folded log-space forward."""

from .plan import Plan, plan_from_torch_circuit  # noqa: F401

__all__ = ["Plan", "plan_from_torch_circuit", "HipCircuit", "HipCircuitStreams", "HipTrainer", "HipCircuitModule", "compile"]


def __getattr__(name):  # lazy: importing the package must not require torch/ROCm
    if name == "HipCircuit":
        from .circuit import HipCircuit

        return HipCircuit
    if name == "HipCircuitStreams":
        from .circuit import HipCircuitStreams

        return HipCircuitStreams
    if name in ("HipTrainer", "HipCircuitModule"):
        from . import training

        return getattr(training, name)
    if name == "compile":
        from .pipeline import compile

        return compile
    raise AttributeError(name)
