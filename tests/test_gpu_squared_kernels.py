"""The launches of a squared circuit's partition function Z on their K = 32 fast paths (round 5):

* `ck_tensordot_lse_fwd_h / _bwd`, `ck_tensordot2_lse_fwd / _bwd` (TorchTensorDotLayer, optimized.py:287-300, under
  ComplexLSESumSemiring / LSESumSemiring, semiring.py:383-408, 441-476; the W / conj W pair of a squared sum layer in one
  launch) with 32 units on every axis take `td32_*` (cirkit_amd/csrc/ck_backward_c.hip): compared with the shape-generic
  kernels (`CK_TD_GENERIC=1`), which the gradient tests of tests/test_training_squared.py pin to the reference's autograd, and
  with a float64 restatement of the layer on the host;
* `ck_param_bmm` on (32, 64) tiles (the Gram matrices W W^T of TorchEinsumParameter, parameters/optimized.py:282-284, and
  their backward) against torch.matmul in float64 for every transposition and ragged sizes.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from cirkit_amd import _capi as capi  # noqa: E402


def _stage_ref(x, w):
    """One TensorDot stage in float64 / complex128: x (F, B, Kj, Kq), w (F, Kk, Kj) -> (F, B, Kq, Kk),
    out[q][k] = log sum_j w[k][j] exp(x[j][q])."""
    m = x.real.amax(dim=2, keepdim=True) if x.is_complex() else x.amax(dim=2, keepdim=True)
    a = torch.exp(x - m)
    y = torch.einsum("fkj,fbjq->fbqk", w.to(a.dtype), a)
    return torch.log(y) + m.transpose(2, 3)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("two", [True, False])
@pytest.mark.parametrize("F,B,H", [(3, 1, 1), (5, 3, 2)])
def test_tensordot_32_units_matches_the_generic_kernels(hip_device, cplx, two, F, B, H):
    K = 32
    g = torch.Generator().manual_seed(17 * F + B + (3 if cplx else 0))
    e = 2 if cplx else 1
    dev = hip_device
    # H blocks per fold in one arena (a Hadamard layer read as a list), as (F * H, B, K * K) values
    blocks = torch.randn(F * H, B, K * K, e, generator=g) * (1.5 if cplx else 1.0)
    w1 = torch.randn(F, K, K, generator=g)
    w2 = torch.randn(F, K, K, generator=g)
    if not cplx:  # (lse-sum: positive weights)
        w1, w2 = w1.abs() + 0.05, w2.abs() + 0.05
    gout = torch.randn(F, B, K * K, e, generator=g)
    ro = (torch.arange(F * H, dtype=torch.int64).reshape(F, H) * (B * K * K)).contiguous()  # in values
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run(generic):
        if generic:
            os.environ["CK_TD_GENERIC"] = "1"
        else:
            os.environ.pop("CK_TD_GENERIC", None)
        try:
            arena, garena = blocks.to(dev).contiguous(), torch.zeros_like(blocks, device=dev)
            mid = torch.zeros(F, B, K * K, e, device=dev)
            out, gmid = torch.zeros_like(mid), torch.zeros_like(mid)
            dw1, dw2 = torch.zeros(F, K, K, device=dev), torch.zeros(F, K, K, device=dev)
            a, b, go, rod = w1.to(dev), w2.to(dev), gout.to(dev), ro.to(dev)
            if two:
                capi.call("ck_tensordot2_lse_fwd", arena.data_ptr(), rod.data_ptr(), H, a.data_ptr(), mid.data_ptr(), b.data_ptr(),
                          out.data_ptr(), F, B, K, K, K, K, int(cplx), stream)
                capi.call("ck_tensordot2_lse_bwd", arena.data_ptr(), garena.data_ptr(), rod.data_ptr(), H, a.data_ptr(), mid.data_ptr(),
                          gmid.data_ptr(), b.data_ptr(), out.data_ptr(), go.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), F, B, K, K, K, K,
                          int(cplx), stream)
            else:
                capi.call("ck_tensordot_lse_fwd_h", arena.data_ptr(), rod.data_ptr(), H, a.data_ptr(), out.data_ptr(), F, B, K, K, K,
                          int(cplx), stream)
                capi.call("ck_tensordot_lse_bwd", arena.data_ptr(), garena.data_ptr(), rod.data_ptr(), H, a.data_ptr(), out.data_ptr(),
                          go.data_ptr(), dw1.data_ptr(), F, B, K, K, K, int(cplx), stream)
            torch.cuda.synchronize()
            return [t.cpu() for t in (out, mid, garena, dw1, dw2)]
        finally:
            os.environ.pop("CK_TD_GENERIC", None)

    fast, slow = run(False), run(True)
    # forward against float64
    x64 = blocks.to(torch.float64)
    x64 = (torch.view_as_complex(x64.contiguous()) if cplx else x64[..., 0]).reshape(F, H, B, K, K).sum(dim=1)
    y1 = _stage_ref(x64, w1.to(torch.float64))
    want = _stage_ref(y1, w2.to(torch.float64)) if two else y1
    got = fast[0].to(torch.float64)
    got = (torch.view_as_complex(got.contiguous()) if cplx else got[..., 0]).reshape(F, B, K, K)
    if cplx:  # (compare exp: the imaginary part of a logarithm is a phase modulo 2 pi)
        d = (torch.exp(got - want) - 1).abs()
    else:
        d = (got - want).abs() / (1 + want.abs())
    assert float(d.max()) <= 2e-4, float(d.max())
    # every output of the fast path against the generic kernels: same sums in the same order, transcendental functions that
    # differ in the last bits -- amplified where signed terms cancel (complex): relative to the block's largest entry
    names = ("out", "mid", "gx", "dw1", "dw2")
    for n, a, b in zip(names, fast, slow):
        if (n in ("mid", "dw2")) and not two:
            continue
        a, b = a.to(torch.float64), b.to(torch.float64)
        if cplx and n in ("out", "mid"):
            ac, bc = torch.view_as_complex(a.contiguous()), torch.view_as_complex(b.contiguous())
            err = float((torch.exp(ac - bc) - 1).abs().max())
            assert err <= 1e-4, (n, err)
        else:
            err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
            assert err <= (2e-4 if cplx else 2e-5), (n, err)
    # (H > 1: every block of the product received the same gradient)
    if H > 1:
        gx = fast[2].reshape(F, H, -1)
        assert torch.equal(gx[:, 0], gx[:, 1])


@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("tb", [0, 1])
@pytest.mark.parametrize("F,M,N,Kd", [(5, 32, 256, 32), (3, 32, 32, 256), (2, 36, 72, 40), (2, 33, 70, 45), (2, 64, 24, 7), (1, 1, 512, 3)])
def test_bmm_tiles(hip_device, ta, tb, F, M, N, Kd):
    g = torch.Generator().manual_seed(M * 131 + N * 7 + Kd + 2 * ta + tb)
    a = torch.randn((F, Kd, M) if ta else (F, M, Kd), generator=g)
    b = torch.randn((F, N, Kd) if tb else (F, Kd, N), generator=g)
    out = torch.full((F, M, N), float("nan"), device=hip_device)
    ad, bd = a.to(hip_device), b.to(hip_device)
    capi.call("ck_param_bmm", ad.data_ptr(), bd.data_ptr(), out.data_ptr(), F, M, N, Kd, ta, tb, 0,
              torch.cuda.current_stream(hip_device).cuda_stream)
    torch.cuda.synchronize()
    a64 = (a.transpose(1, 2) if ta else a).to(torch.float64)
    b64 = (b.transpose(1, 2) if tb else b).to(torch.float64)
    want = a64 @ b64
    err = float((out.cpu().to(torch.float64) - want).abs().max())
    assert err <= 1e-5 * (float(want.abs().max()) + 1.0) * max(1.0, Kd / 32), err


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("F,B,H", [(1, 1, 2), (3, 2, 1)])
def test_tensordot_root_pair_matches_the_generic_kernels(hip_device, cplx, F, B, H):
    """The pair over a SCALAR sum layer (the root of a squared circuit's partition function: 32 x 32 block, one output unit in both
    stages) on `td32_root_*` against the shape-generic kernels and float64."""
    K = 32
    g = torch.Generator().manual_seed(5 * F + B + (11 if cplx else 0))
    e = 2 if cplx else 1
    dev = hip_device
    blocks = torch.randn(F * H, B, K * K, e, generator=g)
    w1, w2 = torch.randn(F, 1, K, generator=g), torch.randn(F, 1, K, generator=g)
    if not cplx:
        w1, w2 = w1.abs() + 0.05, w2.abs() + 0.05
    gout = torch.randn(F, B, 1, e, generator=g)
    ro = (torch.arange(F * H, dtype=torch.int64).reshape(F, H) * (B * K * K)).contiguous()
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run(generic):
        if generic:
            os.environ["CK_TD_GENERIC"] = "1"
        try:
            arena, garena = blocks.to(dev).contiguous(), torch.zeros_like(blocks, device=dev)
            mid, gmid = torch.zeros(F, B, K, e, device=dev), torch.zeros(F, B, K, e, device=dev)
            out = torch.zeros(F, B, 1, e, device=dev)
            dw1, dw2 = torch.zeros(F, 1, K, device=dev), torch.zeros(F, 1, K, device=dev)
            a, b, go, rod = w1.to(dev), w2.to(dev), gout.to(dev), ro.to(dev)
            capi.call("ck_tensordot2_lse_fwd", arena.data_ptr(), rod.data_ptr(), H, a.data_ptr(), mid.data_ptr(), b.data_ptr(), out.data_ptr(),
                      F, B, K, K, 1, 1, int(cplx), stream)
            capi.call("ck_tensordot2_lse_bwd", arena.data_ptr(), garena.data_ptr(), rod.data_ptr(), H, a.data_ptr(), mid.data_ptr(),
                      gmid.data_ptr(), b.data_ptr(), out.data_ptr(), go.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), F, B, K, K, 1, 1, int(cplx),
                      stream)
            torch.cuda.synchronize()
            return [t.cpu() for t in (out, mid, garena, gmid, dw1, dw2)]
        finally:
            os.environ.pop("CK_TD_GENERIC", None)

    fast, slow = run(False), run(True)
    x64 = blocks.to(torch.float64)
    x64 = (torch.view_as_complex(x64.contiguous()) if cplx else x64[..., 0]).reshape(F, H, B, K, K).sum(dim=1)
    y1 = _stage_ref(x64, w1.to(torch.float64))            # (F, B, 32, 1)
    want = _stage_ref(y1, w2.to(torch.float64)).reshape(F, B)
    got = fast[0].to(torch.float64)
    got = (torch.view_as_complex(got.contiguous()) if cplx else got[..., 0]).reshape(F, B)
    d = (torch.exp(got - want) - 1).abs() if cplx else (got - want).abs() / (1 + want.abs())
    assert float(d.max()) <= 2e-4, float(d.max())
    for n, a, b in zip(("out", "mid", "gx", "gmid", "dw1", "dw2"), fast, slow):
        a, b = a.to(torch.float64), b.to(torch.float64)
        if cplx and n in ("out", "mid"):
            err = float((torch.exp(torch.view_as_complex(a.contiguous()) - torch.view_as_complex(b.contiguous())) - 1).abs().max())
            assert err <= 1e-4, (n, err)
        else:
            err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
            assert err <= (2e-4 if cplx else 2e-5), (n, err)


@pytest.mark.parametrize("F,M,Kd", [(4, 32, 256), (2, 64, 128)])
def test_bmm_gram_matrices(hip_device, F, M, Kd):
    """out = a a^T (TorchEinsumParameter of a squared circuit's partition function: W and conj W contracted over the states): both
    operands the SAME tensor -- diagonal tiles read their chunk once -- against float64."""
    g = torch.Generator().manual_seed(F * 7 + M + Kd)
    a = torch.randn(F, M, Kd, generator=g)
    ad = a.to(hip_device)
    out = torch.full((F, M, M), float("nan"), device=hip_device)
    stream = torch.cuda.current_stream(hip_device).cuda_stream
    capi.call("ck_param_bmm", ad.data_ptr(), ad.data_ptr(), out.data_ptr(), F, M, M, Kd, 0, 1, 0, stream)
    capi.call("ck_param_bmm", ad.data_ptr(), ad.data_ptr(), out.data_ptr(), F, M, M, Kd, 0, 1, 1, stream)  # (out += the same)
    torch.cuda.synchronize()
    want = 2 * (a.to(torch.float64) @ a.to(torch.float64).transpose(1, 2))
    err = float((out.cpu().to(torch.float64) - want).abs().max())
    assert err <= 1e-5 * (float(want.abs().max()) + 1.0) * (Kd / 32), err
