import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is ATen on the host.  A GPU box has 256 cores: at ATen's default (one thread per core) the small ops of a
    # folded circuit spend their time in thread hand-overs (the K = 64 gradient checks took 100 s of mostly system time there;
    # bench.py's cpu_baseline probes 8 .. 64 threads for the same reason).  16 threads serve every oracle call of the suite.
    try:
        import torch

        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except ImportError:  # pragma: no cover
        pass


def load_case(name):
    """(plan, tensors as numpy, golden npz dict) of a committed fixture."""
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.plan import Plan

    plan = Plan.load(os.path.join(GOLDEN, name))
    with np.load(os.path.join(GOLDEN, name + "_golden.npz")) as z:
        g = {k: z[k] for k in z.files}
    lit = {k[2:]: g[k] for k in g if k.startswith("w_")}
    tensors = lit if lit else init_plan_tensors(plan)
    return plan, tensors, g


@pytest.fixture(scope="session")
def hip_device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    from cirkit_amd import _capi

    _capi.load()  # must fail loudly on a GPU box if the extension is missing
    return torch.device("cuda:0")
