"""`TorchEinsumParameter` (parameters/optimized.py:282-284) and complex `TorchMatMulParameter` (nodes.py:802-805) on the GPU:
`HipParameter.evaluate` over parameter graphs with einsum / matmul nodes against torch.einsum on the host -- three- and
four-operand patterns, batch and repeated indices, complex operands (what products of more than two circuits and complex
parameters compile to), next to the two-real-matrix patterns that go to `ck_param_bmm`."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from cirkit_amd.parameters import HipParameter, TensorStore  # noqa: E402
from cirkit_amd.plan import IDX_NONE, FoldIndex, ParamGraph, ParamNode  # noqa: E402

LETTERS = "abcdefgh"


def _graph(store, shapes, dtypes, op, config, out_shape, F, g):
    nodes, vals = [], []
    for i, (sh, dt) in enumerate(zip(shapes, dtypes)):
        v = torch.randn((F, *sh), generator=g)
        if dt == "c":
            v = torch.complex(v, torch.randn((F, *sh), generator=g))
        store.set(f"t{i}", v)
        vals.append(v)
        nodes.append(ParamNode("tensor", F, tuple(sh), {"tensor": f"t{i}"}, []))
    nodes.append(ParamNode(op, F, tuple(out_shape), dict(config), [FoldIndex([i], IDX_NONE) for i in range(len(shapes))]))
    return HipParameter(ParamGraph(nodes, FoldIndex([len(nodes) - 1], IDX_NONE), F, tuple(out_shape)), store), vals


@pytest.mark.parametrize("einsum,shapes,dtypes", [
    (((0, 1), (1, 2), (0, 2)), [(3, 4), (4, 5)], "rr"),                      # a matrix product (ck_param_bmm)
    (((0, 1), (2, 3), (0, 2, 1, 3)), [(3, 4), (2, 5)], "rr"),                # an outer product (Kronecker weights)
    (((0, 1), (0, 1), (0,)), [(3, 4), (3, 4)], "rc"),                        # a batch index and a contraction, real x complex
    (((0, 1), (1, 2), (2, 3), (0, 3)), [(3, 4), (4, 2), (2, 5)], "rrr"),     # a chain of three
    (((0, 1), (1, 2), (2, 3), (3, 4), (0, 4)), [(2, 3), (3, 4), (4, 2), (2, 3)], "crcr"),  # four operands, complex
    (((0, 0), (0,)), [(4, 4)], "c"),                                         # a repeated index: the diagonal
    (((0, 1, 2), (2, 1), (0,)), [(3, 4, 5), (5, 4)], "cc"),                  # two contracted indices
])
def test_einsum_parameters(hip_device, einsum, shapes, dtypes):
    g = torch.Generator().manual_seed(len(einsum) * 7 + len(shapes[0]))
    F = 3
    store = TensorStore(hip_device)
    ext = {}
    for e, sh in zip(einsum, shapes):
        ext.update(zip(e, sh))
    out_shape = tuple(ext[i] for i in einsum[-1])
    p, vals = _graph(store, shapes, dtypes, "einsum", {"einsum": einsum}, out_shape, F, g)
    y = p.evaluate(torch.cuda.current_stream(hip_device).cuda_stream)
    torch.cuda.synchronize()
    eq = ",".join("z" + "".join(LETTERS[i] for i in e) for e in einsum[:-1]) + "->z" + "".join(LETTERS[i] for i in einsum[-1])
    cplx = "c" in dtypes
    want = torch.einsum(eq, *[v.to(torch.complex128 if cplx else torch.float64) for v in vals])
    assert tuple(y.shape) == tuple(want.shape) and y.is_complex() == cplx
    assert float((y.cpu().to(want.dtype) - want).abs().max()) <= 2e-5 * (float(want.abs().max()) + 1.0)


def test_complex_matmul_parameter(hip_device):
    g = torch.Generator().manual_seed(3)
    store = TensorStore(hip_device)
    p, (a, b) = _graph(store, [(4, 6), (6, 5)], "cr", "matmul", {}, (4, 5), 2, g)
    y = p.evaluate(torch.cuda.current_stream(hip_device).cuda_stream)
    torch.cuda.synchronize()
    want = torch.matmul(a.to(torch.complex128), b.to(torch.complex128))
    assert float((y.cpu().to(torch.complex128) - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("op,config,shapes,out_shape", [
    ("hadamard", {}, [(3, 5), (3, 5)], (3, 5)),                                  # nodes.py:510-528
    ("kronecker", {}, [(2, 3), (4, 5)], (8, 15)),                                # nodes.py:531-550 (torch.kron per fold)
    ("kronecker", {}, [(3,), (4,)], (12,)),
    ("outer_product", {"dim": 1}, [(3, 2), (3, 4)], (3, 8)),                     # nodes.py:553-612
    ("outer_product", {"dim": 0}, [(2, 5), (3, 5)], (6, 5)),
    ("reduce_sum", {"dim": 1}, [(3, 6)], (3,)),                                  # nodes.py:749-751
    ("reduce_sum", {"dim": 0}, [(4, 2, 3)], (2, 3)),
    ("sum", {}, [(4, 3), (4, 3)], (4, 3)),                                       # nodes.py:491-507
    ("reduce_prod", {"dim": 1}, [(3, 6)], (3,)),                                 # nodes.py:754-756
    ("reduce_lse", {"dim": 0}, [(4, 2, 3)], (2, 3)),                             # nodes.py:759-761
    ("reduce_lse", {"dim": 1}, [(5, 33)], (5,)),
    ("outer_sum", {"dim": 1}, [(3, 2), (3, 4)], (3, 8)),                         # nodes.py:615-653
    ("outer_sum", {"dim": 0}, [(2, 5), (3, 5)], (6, 5)),
    ("index", {"indices": [2, 0, 2, 4], "dim": 0}, [(5, 3)], (4, 3)),            # nodes.py:450-488
    ("gaussian_product_mean", {}, [(4, 1), (4, 1), (3, 1), (3, 1)], (12,)),      # nodes.py:865-908 (mean1, stddev1, mean2, stddev2)
    ("gaussian_product_stddev", {}, [(4, 1), (3, 1)], (12,)),                  # nodes.py:910-938
    ("clamp", {"vmin": -0.3, "vmax": 0.4}, [(5, 7)], (5, 7)),                    # nodes.py:702-728
    ("clamp", {"vmin": 1e-18}, [(5, 7)], (5, 7)),
    ("softplus", {}, [(5, 7)], (5, 7)),                                          # nodes.py:731-739
])
def test_product_sum_and_entrywise_parameter_nodes(hip_device, op, config, shapes, out_shape):
    """The parameter nodes of nodes.py:491-612, 702-751 beside einsum / matmul -- what products of circuits and the clamp / softplus
    activations compile to -- forward against the reference's own formula (the oracle's `eval_param`, fp64) and reverse mode
    (`HipParameter.backward`) against torch autograd through it."""
    from cirkit_amd.plan import ParamGraph as PG
    from oracle.torch_oracle import eval_param

    g = torch.Generator().manual_seed(len(op) * 5 + len(shapes))
    F = 3
    store = TensorStore(hip_device)
    p, vals = _graph(store, shapes, "r" * len(shapes), op, config, out_shape, F, g)
    if op.startswith("gaussian_product"):  # (standard deviations are positive)
        for i, v in enumerate(vals):
            if op.endswith("stddev") or i % 2 == 1:
                vals[i] = v.abs() + 0.3
                store.set(f"t{i}", vals[i])
    if op == "softplus":  # (both sides of the threshold of 20)
        vals[0][0, 0, :3] = torch.tensor([25.0, -30.0, 19.5])
        store.set("t0", vals[0])
    stream = torch.cuda.current_stream(hip_device).cuda_stream
    y = p.evaluate(stream)
    leaves = {f"t{i}": v.double().requires_grad_(True) for i, v in enumerate(vals)}
    torch.set_default_dtype(torch.float64)
    try:
        want = eval_param(p.graph, leaves)
    finally:
        torch.set_default_dtype(torch.float32)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(want.shape) == (F, *out_shape)
    assert float((y.cpu().double() - want.detach()).abs().max()) <= 2e-6 * (float(want.abs().max()) + 1.0)
    dout = torch.randn(want.shape, generator=g)
    want.backward(dout.double())
    grads = {k: torch.zeros_like(store[k]) for k in leaves}
    p.backward(dout.to(hip_device), grads, stream)
    torch.cuda.synchronize()
    for k, v in leaves.items():
        assert float((grads[k].cpu().double() - v.grad).abs().max()) <= 2e-6 * (float(v.grad.abs().max()) + 1.0), (op, k)
