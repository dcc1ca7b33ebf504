"""cirkit_amd/layer_ops.py -- the forward bodies of the TorchLayer subclasses of cirkit_amd/cirkit_plugin.py (row b2) --
on the GPU against the reference's formulas restated in torch (the same restatements oracle/torch_oracle.py uses:
layers/inner.py:126-127,178-187,266-273; layers/optimized.py:89-103,171-178,287-300; layers/input.py:258-266,399-412,
661-670,739-743; semiring.py:383-408)."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _lse_einsum(eq, x, w):
    m = x.real.amax(dim=-1, keepdim=True) if x.is_complex() else x.amax(dim=-1, keepdim=True)
    m = torch.clamp(m, torch.finfo(torch.float32).min, torch.finfo(torch.float32).max)
    e = torch.exp(x - m)
    y = torch.einsum(eq, e, w.to(e.dtype))
    return torch.log(y) + m


def _close(got, want, tol=2e-5):
    got, want = got.cpu(), want.cpu()
    assert got.shape == want.shape and got.dtype == want.dtype
    scale = float(want.abs().max()) + 1.0
    if want.is_complex():
        assert float((got.real - want.real).abs().max()) <= tol * scale
        d = (got.imag - want.imag).abs() % (2 * math.pi)
        assert float(torch.minimum(d, 2 * math.pi - d).max()) <= 1e-3
    else:
        assert float((got - want).abs().max()) <= tol * scale


@pytest.mark.parametrize("F,H,B,Ki,Ko,cplx", [(3, 1, 33, 32, 32, False), (2, 3, 5, 8, 6, False), (2, 2, 7, 16, 4, True)])
def test_sum_cpt_tucker(hip_device, F, H, B, Ki, Ko, cplx):
    from cirkit_amd import _capi as capi
    from cirkit_amd import layer_ops as ops

    g = torch.Generator().manual_seed(F * 100 + Ki)
    x = torch.randn(F, H, B, Ki, generator=g) * 2
    if cplx:
        x = torch.complex(x, torch.randn(F, H, B, Ki, generator=g))
    xd = x.to(hip_device)
    w = torch.rand(F, Ko, H * Ki, generator=g) + 0.05
    _close(ops.sum_lse(xd, w.to(hip_device), capi.CK_SUM_CAT),
           _lse_einsum("fbi,foi->fbo", x.permute(0, 2, 1, 3).flatten(start_dim=2), w))
    w = torch.rand(F, Ko, Ki, generator=g) + 0.05
    _close(ops.sum_lse(xd, w.to(hip_device), capi.CK_SUM_PROD), _lse_einsum("fbi,foi->fbo", x.sum(dim=1), w))
    if H == 2 and not cplx:
        w = torch.rand(F, Ko, Ki * Ki, generator=g) + 0.05
        m0 = x[:, 0].amax(dim=-1, keepdim=True)
        m1 = x[:, 1].amax(dim=-1, keepdim=True)
        y = torch.einsum("fbi,fbj,foij->fbo", torch.exp(x[:, 0] - m0), torch.exp(x[:, 1] - m1), w.view(F, Ko, Ki, Ki))
        _close(ops.sum_lse(xd, w.to(hip_device), capi.CK_SUM_KRON), torch.log(y) + m0 + m1)
    with pytest.raises(ValueError):
        ops.sum_lse(xd, torch.rand(F, Ko, Ki + 1).to(hip_device), capi.CK_SUM_PROD)


@pytest.mark.parametrize("cplx", [False, True])
def test_products_and_tensordot(hip_device, cplx):
    from cirkit_amd import layer_ops as ops

    g = torch.Generator().manual_seed(7)
    F, H, B, K = 3, 3, 9, 4
    x = torch.randn(F, H, B, K, generator=g)
    if cplx:
        x = torch.complex(x, torch.randn(F, H, B, K, generator=g))
    xd = x.to(hip_device)
    _close(ops.hadamard(xd), x.sum(dim=1))
    y0 = x[:, 0]
    for i in range(1, H):
        y0 = torch.flatten(y0.unsqueeze(-1) + x[:, i].unsqueeze(-2), start_dim=-2)
    _close(ops.kronecker(xd), y0)
    Kj, Kq, Kk = 5, 3, 4
    xt = torch.randn(F, 1, B, Kj * Kq, generator=g)
    if cplx:
        xt = torch.complex(xt, torch.randn(F, 1, B, Kj * Kq, generator=g))
    w = torch.rand(F, Kk, Kj, generator=g) + 0.05
    xv = xt.squeeze(1).view(F, B, Kj, Kq).permute(0, 1, 3, 2)
    want = _lse_einsum("fbqj,fkj->fbqk", xv, w).reshape(F, B, Kq * Kk)
    _close(ops.tensordot_lse(xt.to(hip_device), w.to(hip_device), Kj, Kq), want)


def test_input_layers(hip_device):
    from cirkit_amd import layer_ops as ops

    g = torch.Generator().manual_seed(11)
    F, B, K, C = 5, 37, 8, 11
    logits = torch.randn(F, K, C, generator=g)
    x = torch.randint(0, C, (F, B, 1), generator=g)
    idx = torch.arange(F)
    _close(ops.categorical_log_likelihood(x.to(hip_device), logits.to(hip_device)), logits[idx[:, None], :, x.squeeze(2)])
    # a float batch is truncated like x.long() (input.py:400-401); a negative index wraps like torch indexing
    xf = x.to(torch.float32) + 0.25
    _close(ops.categorical_log_likelihood(xf.to(hip_device), logits.to(hip_device)), logits[idx[:, None], :, x.squeeze(2)])
    xn = x.clone()
    xn[0, 0, 0] = -1
    _close(ops.categorical_log_likelihood(xn.to(hip_device), logits.to(hip_device)), logits[idx[:, None], :, xn.squeeze(2)])
    bad = x.clone()
    bad[1, 2, 0] = C
    with pytest.raises(IndexError):  # the reference's advanced indexing raises
        ops.categorical_log_likelihood(bad.to(hip_device), logits.to(hip_device))
    mean, std = torch.randn(F, K, generator=g), torch.rand(F, K, generator=g) + 0.3
    lz = torch.randn(F, K, generator=g)
    xr = torch.randn(F, B, 1, generator=g)
    want = torch.distributions.Normal(mean.unsqueeze(1), std.unsqueeze(1)).log_prob(xr) + lz.unsqueeze(1)
    _close(ops.gaussian_log_likelihood(xr.to(hip_device), mean.to(hip_device), std.to(hip_device), lz.to(hip_device)), want)
    wemb = torch.randn(F, K, C, generator=g)
    sel = wemb[idx[:, None], :, x.squeeze(2)]
    _close(ops.embedding(x.to(hip_device), wemb.to(hip_device), complex_out=True), torch.log(sel.to(torch.complex64)))
    _close(ops.embedding(x.to(hip_device), wemb.abs().to(hip_device), complex_out=False), torch.log(wemb.abs()[idx[:, None], :, x.squeeze(2)]))
    v = torch.rand(F, K, generator=g) + 0.1
    _close(ops.constant_value(v.to(hip_device), 6, log_space=False, complex_out=False), torch.log(v).unsqueeze(1).expand(F, 6, K).contiguous())
    _close(ops.constant_value(v.to(hip_device), 6, log_space=True, complex_out=True), v.unsqueeze(1).expand(F, 6, K).to(torch.complex64).contiguous())


def test_layers_with_more_than_65535_folds(hip_device):
    """Folds ride on grid.y (<= 65535): a layer with more folds -- the leaves of a 256 x 256 image -- is launched in
    chunks.  Categorical, Gaussian, Embedding (complex), dense sum, CP-T (K = 32 tile kernel), Hadamard, Kronecker and
    constant layers with 70 000 folds against plain torch."""
    from cirkit_amd import _capi as capi
    from cirkit_amd import layer_ops as ops

    F, B = 70_000, 3
    g = torch.Generator().manual_seed(7)
    dev = hip_device
    # Categorical
    K, C = 4, 5
    logits = torch.randn(F, K, C, generator=g)
    x = torch.randint(0, C, (F, B, 1), generator=g)
    got = ops.categorical_log_likelihood(x.to(dev), logits.to(dev)).cpu()
    want = logits[torch.arange(F)[:, None], :, x[..., 0]]
    assert torch.equal(got, want)
    # Gaussian
    mean, std = torch.randn(F, K, generator=g), torch.rand(F, K, generator=g) + 0.5
    xf = torch.randn(F, B, 1, generator=g)
    got = ops.gaussian_log_likelihood(xf.to(dev), mean.to(dev), std.to(dev)).cpu()
    want = torch.distributions.Normal(mean[:, None], std[:, None]).log_prob(xf)
    assert float((got - want).abs().max()) <= 1e-4
    # Embedding, complex logarithm
    w = torch.randn(F, K, C, generator=g)
    got = ops.embedding(x.to(dev), w.to(dev), complex_out=True).cpu()
    val = w[torch.arange(F)[:, None], :, x[..., 0]]
    assert float((got.real - val.abs().log()).abs().max()) <= 1e-5 and torch.equal(got.imag > 1.0, val < 0)
    # dense sum (generic kernel) and CP-T on the 32-unit tile kernel
    for Ki, Ko, H, mode in ((4, 3, 1, capi.CK_SUM_CAT), (32, 32, 2, capi.CK_SUM_PROD)):
        xs = torch.randn(F, H, B, Ki, generator=g) - 2
        ws = torch.softmax(torch.randn(F, Ko, Ki, generator=g), dim=-1)
        got = ops.sum_lse(xs.to(dev), ws.to(dev), mode).cpu()
        xin = xs.sum(dim=1) if mode == capi.CK_SUM_PROD else xs[:, 0]
        want = torch.log(torch.einsum("fbi,foi->fbo", torch.exp(xin.double()), ws.double())).float()
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    # products and constants
    xs = torch.randn(F, 2, B, K, generator=g)
    assert float((ops.hadamard(xs.to(dev)).cpu() - xs.sum(dim=1)).abs().max()) <= 1e-6
    kr = ops.kronecker(xs.to(dev)).cpu()
    assert float((kr - (xs[:, 0, :, :, None] + xs[:, 1, :, None, :]).flatten(2)).abs().max()) <= 1e-6
    v = torch.rand(F, K, generator=g) + 0.1
    cv = ops.constant_value(v.to(dev), B, log_space=False, complex_out=False).cpu()
    assert float((cv - v.log()[:, None].expand(F, B, K)).abs().max()) <= 1e-6


def test_no_cpu_fallback():
    from cirkit_amd import layer_ops as ops
    from cirkit_amd._capi import HipExtensionError

    with pytest.raises(HipExtensionError):
        ops.hadamard(torch.zeros(1, 2, 3, 4))


def _grads_close(got, want, tol=2e-4):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape
    scale = float(want.abs().max()) + 1e-6
    assert float((got - want).abs().max()) <= tol * scale, (float((got - want).abs().max()), scale)


@pytest.mark.parametrize("F,H,B,Ki,Ko", [(3, 1, 33, 32, 32), (2, 3, 5, 8, 6), (2, 2, 70, 16, 4), (2, 2, 40, 64, 64)])
def test_sum_layers_backward_is_the_references_autograd(hip_device, F, H, B, Ki, Ko):
    """`loss.backward()` through `layer_ops.sum_lse` (the TorchLayer subclasses of the plugin): d/dx and d/dW from
    ck_sum_lse_bwd equal autograd through the reference's formula (semiring.py:383-408 restated in float64)."""
    from cirkit_amd import _capi as capi
    from cirkit_amd import layer_ops as ops

    g = torch.Generator().manual_seed(F * 131 + Ki)
    x0 = torch.randn(F, H, B, Ki, generator=g) * 2
    gy = torch.randn(F, B, Ko, generator=g)
    cases = [(capi.CK_SUM_CAT, H * Ki, lambda x: x.permute(0, 2, 1, 3).flatten(start_dim=2)),
             (capi.CK_SUM_PROD, Ki, lambda x: x.sum(dim=1))]
    if H == 2 and Ki <= 16:
        cases.append((capi.CK_SUM_KRON, Ki * Ki, lambda x: (x[:, 0].unsqueeze(-1) + x[:, 1].unsqueeze(-2)).flatten(start_dim=-2)))
    for mode, n, prep in cases:
        theta0 = torch.randn(F, Ko, n, generator=g)
        # reference: softmax weights (a parameter graph autograd differentiates on both sides), float64
        xr, tr = x0.double().requires_grad_(True), theta0.double().requires_grad_(True)
        yr = torch.logsumexp(prep(xr).unsqueeze(2) + torch.log_softmax(tr, dim=-1).unsqueeze(1), dim=-1)
        (yr * gy.double()).sum().backward()
        xd, td = x0.to(hip_device).requires_grad_(True), theta0.to(hip_device).requires_grad_(True)
        y = ops.sum_lse(xd, torch.softmax(td, dim=-1), mode)
        assert y.requires_grad
        _close(y.detach(), yr.detach().float())
        (y * gy.to(hip_device)).sum().backward()
        _grads_close(xd.grad, xr.grad)
        _grads_close(td.grad, tr.grad)
    # no graph under no_grad; the complex semiring stays forward-only
    with torch.no_grad():
        assert not ops.sum_lse(x0.to(hip_device).requires_grad_(True), torch.rand(F, Ko, Ki).to(hip_device), capi.CK_SUM_PROD).requires_grad
    if H >= 2:
        with pytest.raises(RuntimeError):  # (complex Kronecker layers stay forward-only)
            ops.kronecker(torch.complex(x0, x0).to(hip_device).requires_grad_(True))


def test_products_and_inputs_backward(hip_device):
    """Hadamard / Kronecker (inner.py:126-127, 178-187), Categorical (input.py:399-412) and Gaussian (input.py:661-670)
    forwards of `layer_ops` under autograd against the same formulas in torch."""
    from cirkit_amd import layer_ops as ops

    g = torch.Generator().manual_seed(23)
    F, H, B, K = 3, 3, 19, 4
    x0 = torch.randn(F, H, B, K, generator=g)
    for kron in (False, True):
        xr = x0.double().requires_grad_(True)
        if kron:
            yr = xr[:, 0]
            for i in range(1, H):
                yr = torch.flatten(yr.unsqueeze(-1) + xr[:, i].unsqueeze(-2), start_dim=-2)
        else:
            yr = xr.sum(dim=1)
        gy = torch.randn(yr.shape, generator=g)
        (yr * gy.double()).sum().backward()
        xd = x0.to(hip_device).requires_grad_(True)
        y = ops.kronecker(xd) if kron else ops.hadamard(xd)
        (y * gy.to(hip_device)).sum().backward()
        _grads_close(xd.grad, xr.grad)
    C = 11
    logits0 = torch.randn(F, K, C, generator=g)
    xc = torch.randint(0, C, (F, B, 1), generator=g)
    idx = torch.arange(F)
    gy = torch.randn(F, B, K, generator=g)
    lr = logits0.double().requires_grad_(True)
    (torch.log_softmax(lr, dim=-1)[idx[:, None], :, xc.squeeze(2)] * gy.double()).sum().backward()
    ld = logits0.to(hip_device).requires_grad_(True)
    y = ops.categorical_log_likelihood(xc.to(hip_device), torch.log_softmax(ld, dim=-1))
    (y * gy.to(hip_device)).sum().backward()
    _grads_close(ld.grad, lr.grad)
    mean0, std0, lz0 = torch.randn(F, K, generator=g), torch.rand(F, K, generator=g) + 0.3, torch.randn(F, K, generator=g)
    xr_ = torch.randn(F, B, 1, generator=g)
    ref = [t.double().requires_grad_(True) for t in (mean0, std0, lz0)]
    yr = torch.distributions.Normal(ref[0].unsqueeze(1), ref[1].unsqueeze(1)).log_prob(xr_.double()) + ref[2].unsqueeze(1)
    (yr * gy.double()).sum().backward()
    dev = [t.to(hip_device).requires_grad_(True) for t in (mean0, std0, lz0)]
    y = ops.gaussian_log_likelihood(xr_.to(hip_device), dev[0], dev[1], dev[2])
    (y * gy.to(hip_device)).sum().backward()
    for a, b in zip(dev, ref):
        _grads_close(a.grad, b.grad)
    # TensorDot (optimized.py:287-300): a dense layer over the rows (b, q) of the permuted input
    Kj, Kq, Kk = 5, 3, 4
    xt0 = torch.randn(F, 1, B, Kj * Kq, generator=g)
    th0 = torch.randn(F, Kk, Kj, generator=g)
    gy = torch.randn(F, B, Kq * Kk, generator=g)
    xr, tr = xt0.double().requires_grad_(True), th0.double().requires_grad_(True)
    xv = xr.squeeze(1).view(F, B, Kj, Kq).permute(0, 1, 3, 2)
    yr = torch.logsumexp(xv.unsqueeze(3) + torch.log_softmax(tr, dim=-1)[:, None, None], dim=-1).reshape(F, B, Kq * Kk)
    (yr * gy.double()).sum().backward()
    xd, td = xt0.to(hip_device).requires_grad_(True), th0.to(hip_device).requires_grad_(True)
    y = ops.tensordot_lse(xd, torch.softmax(td, dim=-1), Kj, Kq)
    _close(y.detach(), yr.detach().float())
    (y * gy.to(hip_device)).sum().backward()
    _grads_close(xd.grad, xr.grad)
    _grads_close(td.grad, tr.grad)
    # ConstantValue (input.py:739-743): log of the value, broadcast over the batch
    v0 = torch.rand(F, K, generator=g) + 0.2
    gy = torch.randn(F, 4, K, generator=g)
    vr = v0.double().requires_grad_(True)
    (torch.log(vr).unsqueeze(1).expand(F, 4, K) * gy.double()).sum().backward()
    vd = v0.to(hip_device).requires_grad_(True)
    (ops.constant_value(vd, 4, log_space=False, complex_out=False) * gy.to(hip_device)).sum().backward()
    _grads_close(vd.grad, vr.grad)
    cg = torch.complex(gy, gy.flip(0))
    vc0 = torch.complex(torch.randn(F, K, generator=g), torch.randn(F, K, generator=g))
    vr = vc0.to(torch.complex128).requires_grad_(True)
    (torch.log(vr).unsqueeze(1).expand(F, 4, K) * cg.to(torch.complex128)).real.sum().backward()
    vd = vc0.to(hip_device).requires_grad_(True)
    (ops.constant_value(vd, 4, log_space=False, complex_out=True) * cg.to(hip_device)).real.sum().backward()
    assert float((vd.grad.cpu().to(torch.complex128) - vr.grad).abs().max()) <= 1e-4 * float(vr.grad.abs().max())


@pytest.mark.parametrize("F,H,B,Ki,Ko,wc", [(2, 2, 7, 16, 4, False), (3, 1, 33, 32, 32, False), (2, 2, 9, 8, 6, True), (2, 3, 5, 4, 3, True),
                                           (2, 2, 1000, 32, 32, False)])
def test_complex_semiring_backward(hip_device, F, H, B, Ki, Ko, wc):
    """Backward through complex-lse-sum (ComplexLSESumSemiring.apply_reduce, semiring.py:441-476; ComplexSafeLog,
    utils.py:22-50): `sum_lse` (three modes, real or complex weights), `hadamard` and `embedding` under autograd against
    autograd through the same formulas in torch complex128, for a real loss Re(sum(y * c))."""
    from cirkit_amd import _capi as capi
    from cirkit_amd import layer_ops as ops

    g = torch.Generator().manual_seed(F * 17 + Ki + (5 if wc else 0))
    x0 = torch.complex(torch.randn(F, H, B, Ki, generator=g), torch.randn(F, H, B, Ki, generator=g) * 3)
    cases = [(capi.CK_SUM_CAT, H * Ki, lambda x: x.permute(0, 2, 1, 3).flatten(start_dim=2)),
             (capi.CK_SUM_PROD, Ki, lambda x: x.sum(dim=1))]
    if H == 2:
        cases.append((capi.CK_SUM_KRON, Ki * Ki, lambda x: (x[:, 0].unsqueeze(-1) + x[:, 1].unsqueeze(-2)).flatten(start_dim=-2)))
    for mode, n, prep in cases:
        w0 = torch.randn(F, Ko, n, generator=g)  # signed weights (sum_weight_activation none, as the squared circuits use)
        if wc:
            w0 = torch.complex(w0, torch.randn(F, Ko, n, generator=g))
        c = torch.complex(torch.randn(F, B, Ko, generator=g), torch.randn(F, B, Ko, generator=g))
        xr = x0.to(torch.complex128).requires_grad_(True)
        wr = w0.to(torch.complex128 if wc else torch.float64).requires_grad_(True)
        v = prep(xr)
        m = v.real.amax(dim=-1, keepdim=True)
        yr = torch.log(torch.einsum("fbn,fon->fbo", torch.exp(v - m), wr.to(torch.complex128))) + m
        (yr * c.to(torch.complex128)).real.sum().backward()
        xd, wd = x0.to(hip_device).requires_grad_(True), w0.to(hip_device).requires_grad_(True)
        y = ops.sum_lse(xd, wd, mode)
        assert y.requires_grad and y.dtype == torch.complex64
        (y * c.to(hip_device)).real.sum().backward()
        scale = float(xr.grad.abs().max())
        assert float((xd.grad.cpu().to(torch.complex128) - xr.grad).abs().max()) <= 5e-4 * scale
        scale = float(wr.grad.abs().max())
        assert wd.grad.dtype == w0.dtype
        assert float((wd.grad.cpu().to(wr.grad.dtype) - wr.grad).abs().max()) <= 5e-4 * scale
    # Hadamard: every child receives the gradient
    c = torch.complex(torch.randn(F, B, Ki, generator=g), torch.randn(F, B, Ki, generator=g))
    xd = x0.to(hip_device).requires_grad_(True)
    (ops.hadamard(xd) * c.to(hip_device)).real.sum().backward()
    xr = x0.clone().requires_grad_(True)
    (xr.sum(dim=1) * c).real.sum().backward()
    assert torch.allclose(xd.grad.cpu(), xr.grad)
    # Embedding: log of a signed real weight
    C = 7
    w0 = torch.randn(F, Ki, C, generator=g)
    xc = torch.randint(0, C, (F, B, 1), generator=g)
    idx = torch.arange(F)
    wr = w0.double().requires_grad_(True)
    (torch.log(wr.to(torch.complex128))[idx[:, None], :, xc.squeeze(2)] * c.to(torch.complex128)).real.sum().backward()
    wd = w0.to(hip_device).requires_grad_(True)
    (ops.embedding(xc.to(hip_device), wd, complex_out=True) * c.to(hip_device)).real.sum().backward()
    _grads_close(wd.grad, wr.grad)
    # Embedding with COMPLEX weights: torch.log of a complex number, gradient gout / conj(w) (utils.py:32-47)
    wc0 = torch.complex(torch.randn(F, Ki, C, generator=g), torch.randn(F, Ki, C, generator=g))
    wcr = wc0.to(torch.complex128).requires_grad_(True)
    ref = torch.log(wcr)[idx[:, None], :, xc.squeeze(2)]
    (ref * c.to(torch.complex128)).real.sum().backward()
    wcd = wc0.to(hip_device).requires_grad_(True)
    got = ops.embedding(xc.to(hip_device), wcd, complex_out=True)
    assert float((got.detach().cpu().to(torch.complex128) - ref.detach()).abs().max()) <= 1e-5
    (got * c.to(hip_device)).real.sum().backward()
    assert float((wcd.grad.cpu().to(torch.complex128) - wcr.grad).abs().max()) <= 1e-4 * float(wcr.grad.abs().max())
