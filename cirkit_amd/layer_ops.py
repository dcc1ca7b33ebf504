"""Layer forwards as plain functions over torch tensors -- the C ABI (`include/cirkit_hip.h`) with the shapes of the
reference's ``TorchLayer.forward`` contracts and already-evaluated parameters.

These are what a layer subclass living INSIDE cirkit calls (`cirkit_amd/cirkit_plugin.py`, SURVEY.md section 8 b2: a
layer compilation rule returning a ``TorchLayer`` subclass): the subclass keeps the reference's parameter graph
(``self.weight()`` ...) and hands the evaluated tensors to the HIP kernel.  Nothing here imports cirkit, so the functions
are exercised on the GPU box (tests/test_gpu_layer_ops.py) where the reference is absent.

Every function enqueues on the current stream of the input's device and raises if the input is not on a ROCm device:
there is no CPU or eager fallback.

Autograd.  Under the real lse-sum semiring `sum_lse`, `hadamard`, `kronecker`, `categorical_log_likelihood` and
`gaussian_log_likelihood` are `torch.autograd.Function`s whose backward is the hand-written kernels of
cirkit_amd/csrc/ck_backward.hip (the ones `HipTrainer` walks a whole plan with): the reference's training loop
(``loss = -circuit(x).mean(); loss.backward(); opt.step()``, notebooks/learning-a-circuit.ipynb) works unchanged on a
circuit compiled with the plugin -- autograd differentiates the reference's own parameter graphs (softmax, ...) and the
gather between layers, these functions supply d/dx and d/dW of each layer.  So do, under complex-lse-sum, `sum_lse` (`ck_sum_lse_bwd_c`), `hadamard`
and `embedding`; TensorDot differentiates as the dense sum layer it is, over a permuted view.  `constant_value` too (its gradient is a batch sum).  Complex Kronecker
forwards record no graph: asking them for gradients raises (`_forward_only`).
"""

from __future__ import annotations

import torch

from . import _capi as capi

__all__ = ["sum_lse", "hadamard", "kronecker", "tensordot_lse", "categorical_log_likelihood", "gaussian_log_likelihood",
           "embedding", "constant_value"]


def _on_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise capi.HipExtensionError(f"{what} is on {t.device}: the HIP layers evaluate on a ROCm device only (no CPU fallback)")


def _forward_only(*tensors: torch.Tensor | None) -> None:
    """The kernels behind these functions write into fresh buffers: their results carry no ``grad_fn``.  Under autograd
    that would silently train nothing (or fail far from the cause), so asking for gradients here raises.  Training on
    the HIP path is `cirkit_amd.training.HipTrainer` (hand-written backward kernels over the whole plan)."""
    if any(t is not None and not t.is_cuda for t in tensors):
        return  # (the device check of the caller raises first: no ROCm device, no forward at all)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise RuntimeError(
            "the HIP layer forwards are inference-only (no autograd graph is recorded): call them under torch.no_grad(), "
            "or train through cirkit_amd.training.HipTrainer")


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _children(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, int, int, int, int]:
    """(F, H, B, Ki) gathered children -> contiguous storage + the (F, H) element offsets the kernels read through."""
    if x.dim() != 4:
        raise ValueError(f"expected an input of shape (F, H, B, Ki), found {tuple(x.shape)}")
    _on_device(x, "the layer input")
    x = x.contiguous()
    F, H, B, Ki = x.shape
    row_off = (torch.arange(F * H, dtype=torch.int64, device=x.device) * (B * Ki)).reshape(F, H)
    return x, row_off, F, H, B, Ki


class _SumLSE(torch.autograd.Function):
    """Real `sum_lse`: forward `ck_sum_lse_fwd`, backward `ck_sum_lse_bwd` (children gradients + dW with float atomics over
    the batch tiles: rounding-level run-to-run differences, as in `HipTrainer`)."""

    @staticmethod
    def forward(ctx, x, weight, row_off, mode):
        F, H, B, Ki = x.shape
        Ko = int(weight.shape[1])
        xf = x.detach().to(torch.float32).contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        out = torch.empty((F, B, Ko), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            capi.call("ck_sum_lse_fwd", xf.data_ptr(), row_off.data_ptr(), w.data_ptr(), out.data_ptr(), F, H, B, Ki, Ko, mode,
                      capi.CK_W_ROWMAJOR, _stream(x.device))
        ctx.save_for_backward(xf, w, out, row_off)
        ctx.mode, ctx.dtypes = mode, (x.dtype, weight.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        xf, w, out, row_off = ctx.saved_tensors
        F, H, B, Ki = xf.shape
        Ko = int(w.shape[1])
        g = gout.to(torch.float32).contiguous()
        gx = torch.empty_like(xf)
        dw = torch.zeros_like(w)
        with torch.cuda.device(xf.device):
            capi.call("ck_sum_lse_bwd", xf.data_ptr(), gx.data_ptr(), row_off.data_ptr(), None, w.data_ptr(), out.data_ptr(),
                      g.data_ptr(), dw.data_ptr(), F, H, B, Ki, Ko, ctx.mode, 0, _stream(xf.device))
        return gx.to(ctx.dtypes[0]), dw.to(ctx.dtypes[1]), None, None


class _Product(torch.autograd.Function):
    """Real `hadamard` / `kronecker` (log space: sums of the children's values)."""

    @staticmethod
    def forward(ctx, x, row_off, kron):
        F, H, B, K = x.shape
        cplx = x.is_complex()
        xf = x.detach().to(torch.complex64 if cplx else torch.float32).contiguous()
        out = torch.empty((F, B, K**H if kron else K), dtype=xf.dtype, device=x.device)
        with torch.cuda.device(x.device):
            capi.call("ck_kronecker_fwd" if kron else "ck_hadamard_fwd", xf.data_ptr(), row_off.data_ptr(), out.data_ptr(), F, H,
                      B, K, 2 if cplx else 1, _stream(x.device))
        ctx.save_for_backward(row_off)
        ctx.kron, ctx.shape, ctx.dtype = kron, tuple(x.shape), x.dtype
        return out if x.dtype == out.dtype else out.to(x.dtype)

    @staticmethod
    def backward(ctx, gout):
        (row_off,) = ctx.saved_tensors
        F, H, B, K = ctx.shape
        cplx = gout.is_complex()
        g = gout.to(torch.complex64 if cplx else torch.float32).contiguous()
        gx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
        with torch.cuda.device(g.device):
            if cplx:  # a complex (F, H, B, K) block is a float block of 2 K units per row: every child receives gout
                capi.call("ck_hadamard_bwd", gx.data_ptr(), (row_off * 2).data_ptr(), g.data_ptr(), F, H, B, 2 * K, 0, _stream(g.device))
            else:
                capi.call("ck_kronecker_bwd" if ctx.kron else "ck_hadamard_bwd", gx.data_ptr(), row_off.data_ptr(), g.data_ptr(), F,
                          H, B, K, 0, _stream(g.device))
        return gx.to(ctx.dtype), None, None


class _SumCLSE(torch.autograd.Function):
    """Complex `sum_lse` (ComplexLSESumSemiring.apply_reduce, semiring.py:441-476): forward `ck_sum_lse_fwd_c`, backward
    `ck_sum_lse_bwd_c` -- y is holomorphic in inputs and weights, gradients are conj(dy/d.) * gout as torch defines them;
    real weights receive the real part."""

    @staticmethod
    def forward(ctx, x, weight, row_off, mode):
        F, H, B, Ki = x.shape
        Ko = int(weight.shape[1])
        xc = x.detach().to(torch.complex64).contiguous()
        w = weight.detach().contiguous()
        w = w.to(torch.complex64) if w.is_complex() else w.to(torch.float32)
        out = torch.empty((F, B, Ko), dtype=torch.complex64, device=x.device)
        with torch.cuda.device(x.device):
            capi.call("ck_sum_lse_fwd_c", xc.data_ptr(), row_off.data_ptr(), w.data_ptr(), out.data_ptr(), F, H, B, Ki, Ko,
                      mode, 1 if w.is_complex() else 0, _stream(x.device))
        ctx.save_for_backward(xc, w, out, row_off)
        ctx.mode, ctx.dtypes = mode, (x.dtype, weight.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        xc, w, out, row_off = ctx.saved_tensors
        F, H, B, Ki = xc.shape
        Ko = int(w.shape[1])
        g = gout.to(torch.complex64).contiguous()
        gx = torch.empty_like(xc)
        dw = torch.zeros_like(w)
        with torch.cuda.device(xc.device):
            capi.call("ck_sum_lse_bwd_c", xc.data_ptr(), gx.data_ptr(), row_off.data_ptr(), w.data_ptr(), out.data_ptr(),
                      g.data_ptr(), dw.data_ptr(), F, H, B, Ki, Ko, ctx.mode, 1 if w.is_complex() else 0, _stream(xc.device))
        return gx.to(ctx.dtypes[0]), dw.to(ctx.dtypes[1]), None, None


class _Embedding(torch.autograd.Function):
    """`embedding`: out[f, b, :] = log(weight[f, :, x[f, b]]) (ComplexSafeLog of a real number under complex-lse-sum,
    utils.py:22-50); d out / d weight = 1 / weight, so the gradient is the scatter-add of Re(gout) over the batch
    (`ck_categorical_bwd`) divided by the weight."""

    @staticmethod
    def forward(ctx, weight, xi, complex_out):
        F, K, C = weight.shape
        B = xi.shape[1]
        dev = xi.device
        ctx.cplx_w = bool(weight.is_complex())
        if ctx.cplx_w:  # complex weights: torch.log of a complex number, (log|w|, arg w) (utils.py:32-35)
            table = torch.zeros((F, C + 1, K), dtype=torch.complex64, device=dev)
            table[:, :C] = weight.detach().to(torch.complex64).transpose(1, 2)
            scope = torch.arange(F, dtype=torch.int64, device=dev)
            out = torch.empty((F, B, K), dtype=torch.complex64, device=dev)
            with torch.cuda.device(dev):
                capi.call("ck_embedding_clog_c_fwd", table.data_ptr(), xi.data_ptr(), scope.data_ptr(), out.data_ptr(), F, B, K, C, F,
                          _stream(dev))
            ctx.save_for_backward(xi, scope, table)
            ctx.dims, ctx.dtype = (F, K, C, B), weight.dtype
            return out
        # (F, C + 1, K) like every gather table (row C: the integral row of the marginal queries, not reachable from here)
        table = torch.zeros((F, C + 1, K), dtype=torch.float32, device=dev)
        table[:, :C] = weight.detach().to(torch.float32).transpose(1, 2)
        scope = torch.arange(F, dtype=torch.int64, device=dev)
        out = torch.empty((F, B, K), dtype=torch.complex64 if complex_out else torch.float32, device=dev)
        with torch.cuda.device(dev):
            capi.call("ck_embedding_clog_fwd" if complex_out else "ck_embedding_log_fwd", table.data_ptr(), xi.data_ptr(),
                      scope.data_ptr(), out.data_ptr(), F, B, K, C, F, _stream(dev))
        ctx.save_for_backward(xi, scope, table)
        ctx.dims, ctx.dtype = (F, K, C, B), weight.dtype
        return out

    @staticmethod
    def backward(ctx, gout):
        xi, scope, table = ctx.saved_tensors
        F, K, C, B = ctx.dims
        if ctx.cplx_w:  # ComplexSafeLog.backward (utils.py:44-47): gout / conj(w), scatter-added over the batch as float pairs
            g = torch.view_as_real(gout.to(torch.complex64).contiguous()).reshape(F, B, 2 * K)
            dtable = torch.zeros((F, C + 1, 2 * K), dtype=torch.float32, device=g.device)
            with torch.cuda.device(g.device):
                capi.call("ck_categorical_bwd", g.data_ptr(), None, xi.data_ptr(), scope.data_ptr(), dtable.data_ptr(), F, B, 2 * K, C, 1,
                          None, _stream(g.device))
            num = torch.view_as_complex(dtable[:, :C].reshape(F, C, K, 2).contiguous())
            dw = torch.where(num != 0, num / table[:, :C].conj(), torch.zeros_like(num))
            return dw.transpose(1, 2).contiguous().to(ctx.dtype), None, None
        g = (gout.real if gout.is_complex() else gout).to(torch.float32).contiguous()
        dtable = torch.zeros((F, C + 1, K), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            capi.call("ck_categorical_bwd", g.data_ptr(), None, xi.data_ptr(), scope.data_ptr(), dtable.data_ptr(), F, B, K, C, 1, None,
                      _stream(g.device))
        # d log w / d w = 1 / w at the entries some batch row selected; the others have no gradient (also where w == 0:
        # 0 / 0 would be NaN there and poison the optimizer's moments -- the reference's indexing backward leaves 0)
        num = dtable[:, :C]
        dw = torch.where(num != 0, num / table[:, :C], torch.zeros_like(num))
        return dw.transpose(1, 2).contiguous().to(ctx.dtype), None, None


class _Categorical(torch.autograd.Function):
    """`categorical_log_likelihood`: the gather of table rows; backward = `ck_categorical_bwd`, a scatter-add into the
    (F, C + 1, K) table, handed back in the (F, K, C) shape of the logits."""

    @staticmethod
    def forward(ctx, logits, xi):
        F, K, C = logits.shape
        B = xi.shape[1]
        dev = xi.device
        # the gather kernel reads a (F, C + 1, K) table (row C = the integral row, unused here) through per-fold variables
        table = torch.empty((F, C + 1, K), dtype=torch.float32, device=dev)
        table[:, :C] = logits.detach().to(torch.float32).transpose(1, 2)
        table[:, C] = 0.0
        scope = torch.arange(F, dtype=torch.int64, device=dev)
        out = torch.empty((F, B, K), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            capi.call("ck_categorical_fwd", table.data_ptr(), xi.data_ptr(), scope.data_ptr(), out.data_ptr(), F, B, K, C, F,
                      _stream(dev))
        ctx.save_for_backward(xi, scope)
        ctx.dims, ctx.dtype = (F, K, C, B), logits.dtype
        return out

    @staticmethod
    def backward(ctx, gout):
        xi, scope = ctx.saved_tensors
        F, K, C, B = ctx.dims
        g = gout.to(torch.float32).contiguous()
        dtable = torch.zeros((F, C + 1, K), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            capi.call("ck_categorical_bwd", g.data_ptr(), None, xi.data_ptr(), scope.data_ptr(), dtable.data_ptr(), F, B, K, C, 1, None,
                      _stream(g.device))
        return dtable[:, :C].transpose(1, 2).contiguous().to(ctx.dtype), None


class _Gaussian(torch.autograd.Function):
    """`gaussian_log_likelihood`; backward = `ck_gaussian_bwd` (d mean, d stddev as batch sums) and the batch sum of the
    incoming gradient for the log-partition."""

    @staticmethod
    def forward(ctx, mean, stddev, log_partition, xt):
        F, K = mean.shape
        B = xt.shape[1]
        dev = xt.device
        m = mean.detach().to(torch.float32).contiguous()
        sd = stddev.detach().to(torch.float32).contiguous()
        lz = None if log_partition is None else log_partition.detach().to(torch.float32).contiguous()
        scope = torch.arange(F, dtype=torch.int64, device=dev)
        out = torch.empty((F, B, K), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            capi.call("ck_gaussian_fwd", m.data_ptr(), sd.data_ptr(), None if lz is None else lz.data_ptr(), xt.data_ptr(),
                      scope.data_ptr(), out.data_ptr(), F, B, K, F, _stream(dev))
        ctx.save_for_backward(m, sd, xt, scope)
        ctx.has_lz = log_partition is not None
        ctx.dtypes = (mean.dtype, stddev.dtype, None if log_partition is None else log_partition.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        m, sd, xt, scope = ctx.saved_tensors
        F, K = m.shape
        B = xt.shape[1]
        g = gout.to(torch.float32).contiguous()
        dm, ds = torch.zeros_like(m), torch.zeros_like(sd)
        with torch.cuda.device(g.device):
            capi.call("ck_gaussian_bwd", g.data_ptr(), xt.data_ptr(), scope.data_ptr(), m.data_ptr(), sd.data_ptr(), dm.data_ptr(),
                      ds.data_ptr(), F, B, K, _stream(g.device))
        dlz = g.sum(dim=1).to(ctx.dtypes[2]) if ctx.has_lz else None
        return dm.to(ctx.dtypes[0]), ds.to(ctx.dtypes[1]), dlz, None



def sum_lse(x: torch.Tensor, weight: torch.Tensor, mode: int = capi.CK_SUM_CAT) -> torch.Tensor:
    """``TorchSumLayer.forward`` (inner.py:266-273, mode CK_SUM_CAT), ``TorchCPTLayer.forward`` (optimized.py:171-178,
    CK_SUM_PROD) and ``TorchTuckerLayer.forward`` (optimized.py:89-103, CK_SUM_KRON) under lse-sum / complex-lse-sum:
    x (F, H, B, Ki) log-space children, weight (F, Ko, N) linear-space -> (F, B, Ko)."""
    x, row_off, F, H, B, Ki = _children(x)
    if weight.dim() != 3 or weight.shape[0] != F:
        raise ValueError(f"expected a weight of shape (F={F}, Ko, N), found {tuple(weight.shape)}")
    _on_device(weight, "the weight")
    Ko = int(weight.shape[1])
    n_in = {capi.CK_SUM_CAT: H * Ki, capi.CK_SUM_PROD: Ki, capi.CK_SUM_KRON: Ki**H}[mode]
    if weight.shape[2] != n_in:
        raise ValueError(f"the weight has {weight.shape[2]} inputs per output, the layer contracts {n_in}")
    with torch.cuda.device(x.device):
        st = _stream(x.device)
        if x.is_complex():
            return _SumCLSE.apply(x, weight, row_off, mode)
        if weight.is_complex():
            raise ValueError("complex weights under the real lse-sum semiring")
        return _SumLSE.apply(x, weight, row_off, mode)


def hadamard(x: torch.Tensor) -> torch.Tensor:
    """``TorchHadamardLayer.forward`` (inner.py:126-127) in log space: the sum over the arity axis."""
    x, row_off, F, H, B, K = _children(x)
    return _Product.apply(x, row_off, False)


def kronecker(x: torch.Tensor) -> torch.Tensor:
    """``TorchKroneckerLayer.forward`` (inner.py:178-187), any arity: (F, H, B, K) -> (F, B, K ** H)."""
    if x.is_complex():
        _forward_only(x)
    x, row_off, F, H, B, K = _children(x)
    if H < 2:
        raise ValueError("The arity should be at least 2")
    if not x.is_complex():
        return _Product.apply(x, row_off, True)
    out = torch.empty((F, B, K**H), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        capi.call("ck_kronecker_fwd", x.data_ptr(), row_off.data_ptr(), out.data_ptr(), F, H, B, K, 2, _stream(x.device))
    return out


def tensordot_lse(x: torch.Tensor, weight: torch.Tensor, num_contract_units: int, num_batch_units: int) -> torch.Tensor:
    """``TorchTensorDotLayer.forward`` (optimized.py:287-300): x (F, 1, B, Kj * Kq), weight (F, Kk, Kj) -> (F, B, Kq * Kk)."""
    x, row_off, F, H, B, Ki = _children(x)
    Kj, Kq = int(num_contract_units), int(num_batch_units)
    if H != 1 or Ki != Kj * Kq or weight.shape[0] != F or weight.shape[2] != Kj:
        raise ValueError(f"tensordot: input {tuple(x.shape)}, weight {tuple(weight.shape)}, Kj={Kj}, Kq={Kq}")
    _on_device(weight, "the weight")
    Kk = int(weight.shape[1])
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        # under autograd: the layer IS a dense sum layer over the rows (b, q) of the permuted input (optimized.py:289-296);
        # the permutation is torch's (it knows its own backward), the contraction and its backward are `_SumLSE` / `_SumCLSE`
        xp = x.view(F, B, Kj, Kq).permute(0, 1, 3, 2).reshape(F, 1, B * Kq, Kj)
        ro = (torch.arange(F, dtype=torch.int64, device=x.device) * (B * Kq * Kj)).reshape(F, 1)
        fn = _SumCLSE if x.is_complex() else _SumLSE
        if weight.is_complex() and not x.is_complex():
            raise ValueError("complex weights under the real lse-sum semiring")
        return fn.apply(xp, weight, ro, capi.CK_SUM_PROD).view(F, B, Kq * Kk)
    with torch.cuda.device(x.device):
        st = _stream(x.device)
        if x.is_complex():
            w = weight.contiguous()
            w = w.to(torch.complex64) if w.is_complex() else w.to(torch.float32)
            out = torch.empty((F, B, Kq * Kk), dtype=torch.complex64, device=x.device)
            capi.call("ck_tensordot_lse_fwd_c", x.data_ptr(), row_off.data_ptr(), w.data_ptr(), out.data_ptr(), F, B, Kj, Kq, Kk,
                      1 if w.is_complex() else 0, st)
            return out
        w = weight.to(torch.float32).contiguous()
        out = torch.empty((F, B, Kq * Kk), dtype=torch.float32, device=x.device)
        capi.call("ck_tensordot_lse_fwd", x.data_ptr(), row_off.data_ptr(), w.data_ptr(), out.data_ptr(), F, B, Kj, Kq, Kk, st)
        return out


def _discrete_input(x: torch.Tensor, num_states: int) -> torch.Tensor:
    """(F, B, 1) -> (F, B) int32; like the reference's advanced indexing an out-of-range value is an IndexError."""
    _on_device(x, "the layer input")
    if x.dim() != 3 or x.shape[2] != 1:
        raise ValueError(f"expected an input of shape (F, B, 1), found {tuple(x.shape)}")
    if x.is_floating_point():
        x = x.long()  # input.py:400-401
    x = x.squeeze(dim=2)
    if x.numel():
        lo, hi = torch.stack(torch.aminmax(x)).tolist()  # ONE host synchronisation for both bounds
        if lo < -num_states or hi >= num_states:
            raise IndexError(f"index out of range for {num_states} states")
    x = torch.where(x < 0, x + num_states, x)  # torch indexing wraps negative indices
    return x.to(torch.int32).contiguous()


def categorical_log_likelihood(x: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """``TorchCategoricalLayer.log_unnormalized_likelihood`` (input.py:399-412): x (F, B, 1) categories, logits (F, K, C)
    (``log(probs())`` or ``logits()``) -> (F, B, K)."""
    F, K, C = logits.shape
    xi = _discrete_input(x, C)
    return _Categorical.apply(logits, xi)


def gaussian_log_likelihood(x: torch.Tensor, mean: torch.Tensor, stddev: torch.Tensor,
                            log_partition: torch.Tensor | None = None) -> torch.Tensor:
    """``TorchGaussianLayer.log_unnormalized_likelihood`` (input.py:661-670): x (F, B, 1), mean / stddev (F, K)."""
    _forward_only(x)  # (no gradient with respect to the data)
    _on_device(x, "the layer input")
    if x.dim() != 3 or x.shape[2] != 1:
        raise ValueError(f"expected an input of shape (F, B, 1), found {tuple(x.shape)}")
    xt = x.detach().squeeze(dim=2).to(torch.float32).contiguous()  # (F, B): "variable" f of a (D = F, B) staging copy
    return _Gaussian.apply(mean, stddev, log_partition, xt)


def embedding(x: torch.Tensor, weight: torch.Tensor, *, complex_out: bool) -> torch.Tensor:
    """``TorchEmbeddingLayer.forward`` (input.py:258-266) mapped into lse-sum (log) / complex-lse-sum (complex log):
    x (F, B, 1) states, weight (F, K, C) real -- or complex under complex-lse-sum (rules/parameters.py:75-86) -> (F, B, K)."""
    if weight.is_complex() and not complex_out:
        raise ValueError("embedding: complex weights under the real lse-sum semiring")
    F, K, C = weight.shape
    xi = _discrete_input(x, C)
    return _Embedding.apply(weight, xi, bool(complex_out))


def constant_value(value: torch.Tensor, batch_size: int, *, log_space: bool, complex_out: bool) -> torch.Tensor:
    """``TorchConstantValueLayer.forward`` (input.py:739-743): value (F, K) broadcast over the batch."""
    _on_device(value, "the value")
    if value.is_complex() and not complex_out:
        raise ValueError("complex constant value under the real lse-sum semiring")
    return _Constant.apply(value, int(batch_size), bool(log_space), bool(complex_out))


class _Constant(torch.autograd.Function):
    """`constant_value`: out[f, b, :] = value[f] (log_space) or log(value[f]); the gradient is the batch sum of the incoming
    one (a handful of rows: these layers carry the partition function of a squared circuit, batch 1), times conj(1 / value)
    for the logarithm; a real value receives the real part."""

    @staticmethod
    def forward(ctx, value, batch_size, log_space, complex_out):
        F, K = value.shape
        v = value.detach().contiguous()
        v = v.to(torch.complex64) if v.is_complex() else v.to(torch.float32)
        out = torch.empty((F, batch_size, K), dtype=torch.complex64 if complex_out else torch.float32, device=value.device)
        with torch.cuda.device(value.device):
            capi.call("ck_constant_fwd", v.data_ptr(), out.data_ptr(), F, batch_size, K, 1 if log_space else 0,
                      1 if v.is_complex() else 0, 1 if complex_out else 0, _stream(value.device))
        ctx.save_for_backward(v)
        ctx.log_space, ctx.dtype = log_space, value.dtype
        return out

    @staticmethod
    def backward(ctx, gout):
        (v,) = ctx.saved_tensors
        g = gout.sum(dim=1)
        if not ctx.log_space:
            g = g / (v.conj() if v.is_complex() else v)
        if not v.is_complex() and g.is_complex():
            g = g.real
        return g.to(ctx.dtype), None, None, None
