"""CPU ORACLE -- test infrastructure, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; nothing under ``cirkit_amd/`` does (tests/test_no_oracle_in_product.py enforces it).

What it is: an op-for-op restatement, in plain CPU PyTorch, of the reference's folded log-space
forward (`cirkit.backend.torch`), driven by a `cirkit_amd.plan.Plan` instead of by the reference's
Python objects.  The arithmetic library of the reference for this path *is* PyTorch ATen
(``torch>=2.3`` in the reference's pyproject.toml:36; 2.10.0 in this image), so the same ATen calls
in the same order reproduce the reference bit-for-bit on CPU; that is pinned by

* tests/golden/*.npz -- outputs of the real reference, generated in the build container by
  tests/golden/make_fixtures.py (which imports /root/reference), and
* the reference's own known-answer tests (tests/symbolic/test_utils.py:293-503 of the reference),
  restated in tests/test_oracle_golden.py::test_reference_known_answers (fixtures tests/golden/kat_*).

Every function cites the reference lines it follows (paths relative to the reference checkout).
"""

from __future__ import annotations

import functools
import itertools
import math
from typing import Mapping

import numpy as np
import torch
from torch import Tensor

from cirkit_amd.plan import (
    IDX_ARRAY,
    IDX_NONE,
    IDX_UNSQ0,
    IDX_UNSQ1,
    FoldIndex,
    LayerSpec,
    ParamGraph,
    Plan,
)


# --------------------------------------------------------------------------------------------
# semirings -- cirkit/backend/torch/semiring.py
# --------------------------------------------------------------------------------------------
def _csafelog(x: Tensor) -> Tensor:
    # utils.py:32-50: the forward of ComplexSafeLog is plain torch.log
    return torch.log(x)


class _LSE:
    name = "lse-sum"

    @staticmethod
    def cast(x: Tensor) -> Tensor:  # semiring.py:358-364
        if x.is_floating_point():
            return x
        return x.to(torch.get_default_dtype())

    @staticmethod
    def prod(x: Tensor, dim: int) -> Tensor:  # semiring.py:375-376
        return x.sum(dim=dim)

    @staticmethod
    def mul(*xs: Tensor) -> Tensor:  # semiring.py:379-380
        return functools.reduce(torch.add, xs)

    @staticmethod
    def real(x: Tensor) -> Tensor:
        return x

    @staticmethod
    def log(x: Tensor) -> Tensor:
        return torch.log(x)

    @staticmethod
    def from_sum_product(x: Tensor) -> Tensor:  # semiring.py:495-497
        return torch.log(x)

    @staticmethod
    def from_lse(x: Tensor) -> Tensor:
        return x


class _CLSE(_LSE):
    name = "complex-lse-sum"

    @staticmethod
    def cast(x: Tensor) -> Tensor:  # semiring.py:416-422
        if x.is_complex():
            return x
        if x.is_floating_point():
            return x.to(x.dtype.to_complex())
        return x.to(torch.get_default_dtype().to_complex())

    @staticmethod
    def real(x: Tensor) -> Tensor:
        return x.real

    @staticmethod
    def log(x: Tensor) -> Tensor:
        return _csafelog(x)

    @staticmethod
    def from_sum_product(x: Tensor) -> Tensor:  # semiring.py:507-509
        return _csafelog(_CLSE.cast(x))

    @staticmethod
    def from_lse(x: Tensor) -> Tensor:  # semiring.py:512-514
        return _CLSE.cast(x)


def _apply_reduce(sr, func, *xs: Tensor, dim: int, keepdim: bool) -> Tensor:
    """semiring.py:383-408 (lse-sum) and :441-476 (complex-lse-sum): subtract the max of the INPUTS
    only, evaluate the contraction in linear space, log, add the maxima back."""
    max_xs = [
        torch.clamp(
            torch.amax(sr.real(xi), dim=dim, keepdim=True),
            min=torch.finfo(sr.real(xi).dtype).min,
            max=torch.finfo(sr.real(xi).dtype).max,
        )
        for xi in xs
    ]
    exp_xs = [torch.exp(xi - m) for xi, m in zip(xs, max_xs)]
    y = func(*exp_xs)
    red = functools.reduce(torch.add, max_xs)
    if not keepdim:
        red = red.squeeze(dim)
    return sr.log(y) + red


def _einsum(sr, equation: str, *, inputs, operands, dim: int, keepdim: bool) -> Tensor:
    """SemiringImpl.einsum, semiring.py:147-202 (string-equation branch)."""
    operands = tuple(sr.cast(o) for o in operands)

    def func(*exp_xs: Tensor) -> Tensor:
        return torch.einsum(equation, *exp_xs, *operands)

    return _apply_reduce(sr, func, *inputs, dim=dim, keepdim=keepdim)


# --------------------------------------------------------------------------------------------
# parameter graphs -- cirkit/backend/torch/parameters/{parameter,nodes,optimized}.py
# --------------------------------------------------------------------------------------------
def _select(outputs: list[Tensor], fi: FoldIndex) -> Tensor:
    """parameter.py:41-47 / circuits.py:42-47: cat the producers along folds, then index."""
    t = outputs[fi.ids[0]] if len(fi.ids) == 1 else torch.cat([outputs[i] for i in fi.ids], dim=0)
    if fi.kind == IDX_ARRAY:
        return t[torch.from_numpy(np.asarray(fi.array))]
    if fi.kind == IDX_UNSQ0:
        return t[None]
    if fi.kind == IDX_UNSQ1:
        return t[:, None]
    assert fi.kind == IDX_NONE
    return t


def eval_param(pg: ParamGraph, tensors: Mapping[str, Tensor]) -> Tensor:
    """TorchParameter.forward -> evaluate, parameter.py:180-188; node forwards in nodes.py."""
    outs: list[Tensor] = []
    for n in pg.nodes:
        xs = [_select(outs, fi) for fi in n.inputs]
        c = n.config
        if n.op == "tensor":  # nodes.py:203-220
            y = tensors[c["tensor"]]
        elif n.op == "pointer":  # nodes.py:277-279
            y = tensors[c["tensor"]]
            if c.get("fold_idx") is not None:
                y = y[torch.tensor(c["fold_idx"])]
        elif n.op == "softmax":  # nodes.py:771-772
            y = torch.softmax(xs[0], dim=c["dim"] + 1)
        elif n.op == "log_softmax":  # nodes.py:782-783
            y = torch.log_softmax(xs[0], dim=c["dim"] + 1)
        elif n.op == "sigmoid":  # nodes.py:678-679
            y = torch.sigmoid(xs[0])
        elif n.op == "scaled_sigmoid":  # nodes.py:698-699
            y = torch.sigmoid(xs[0]) * (c["vmax"] - c["vmin"]) + c["vmin"]
        elif n.op == "exp":
            y = torch.exp(xs[0])
        elif n.op == "log":
            y = torch.log(xs[0])
        elif n.op == "square":
            y = torch.square(xs[0])
        elif n.op == "sum":  # nodes.py:506-507
            y = xs[0] + xs[1]
        elif n.op == "hadamard":  # nodes.py:527-528
            y = xs[0] * xs[1]
        elif n.op == "kronecker":  # nodes.py:549-550
            y = torch.vmap(torch.kron)(xs[0], xs[1])
        elif n.op == "outer_product":  # nodes.py:604-612
            d = c["dim"]
            y = (xs[0].unsqueeze(d + 2) * xs[1].unsqueeze(d + 1)).reshape(n.num_folds, *n.shape)
        elif n.op == "reduce_sum":  # nodes.py:750-751
            y = torch.sum(xs[0], dim=c["dim"] + 1)
        elif n.op == "reduce_prod":  # nodes.py:755-756
            y = torch.prod(xs[0], dim=c["dim"] + 1)
        elif n.op == "reduce_lse":  # nodes.py:760-761
            y = torch.logsumexp(xs[0], dim=c["dim"] + 1)
        elif n.op == "outer_sum":  # nodes.py:646-653
            d = c["dim"]
            y = (xs[0].unsqueeze(d + 2) + xs[1].unsqueeze(d + 1)).reshape(n.num_folds, *n.shape)
        elif n.op == "index":  # nodes.py:487-488 (the FIRST axis of the per-fold value, whatever `dim` says: as the reference does)
            y = xs[0][:, torch.tensor(c["indices"])]
        elif n.op == "gaussian_product_mean":  # nodes.py:899-908
            mean1, stddev1, mean2, stddev2 = xs
            var1, var2 = torch.square(stddev1), torch.square(stddev2)
            inv_var12 = torch.reciprocal(var1.unsqueeze(dim=2) + var2.unsqueeze(dim=1))
            wm1 = mean1.unsqueeze(dim=2) * var2.unsqueeze(dim=1)
            wm2 = mean2.unsqueeze(dim=1) * var1.unsqueeze(dim=2)
            y = ((wm1 + wm2) * inv_var12).view(-1, *n.shape)
        elif n.op == "gaussian_product_stddev":  # nodes.py:931-938
            var1, var2 = torch.square(xs[0]), torch.square(xs[1])
            inv_var1, inv_var2 = torch.reciprocal(var1).unsqueeze(dim=2), torch.reciprocal(var2).unsqueeze(dim=1)
            y = torch.sqrt(torch.reciprocal(inv_var1 + inv_var2)).view(-1, *n.shape)
        elif n.op == "clamp":  # nodes.py:727-728
            y = torch.clamp(xs[0], min=c.get("vmin"), max=c.get("vmax"))
        elif n.op == "softplus":  # nodes.py:738-739
            y = torch.nn.functional.softplus(xs[0])
        elif n.op == "conj":  # nodes.py:745-746
            y = torch.conj(xs[0])
        elif n.op == "mixing_weight":  # nodes.py:857-862
            d = torch.vmap(torch.vmap(torch.diag, in_dims=1))(xs[0])
            y = d.permute(0, 2, 1, 3).flatten(start_dim=2)
        elif n.op == "matmul":  # nodes.py:802-805
            y = torch.matmul(xs[0], xs[1])
        elif n.op == "einsum":  # optimized.py:282-284 (TorchEinsumParameter.forward)
            folded = [(0,) + tuple(i + 1 for i in e) for e in c["einsum"]]
            args = tuple(itertools.chain.from_iterable(zip(xs, folded[:-1])))
            y = torch.einsum(*args, folded[-1])
        elif n.op == "flatten":  # nodes.py:843-844
            y = torch.flatten(xs[0], start_dim=c["start_dim"] + 1, end_dim=c["end_dim"] + 1)
        elif n.op == "gaussian_product_log_partition":  # nodes.py:975-988
            mean1, stddev1, mean2, stddev2 = xs
            var12 = torch.square(stddev1).unsqueeze(dim=2) + torch.square(stddev2).unsqueeze(dim=1)
            sq = torch.square(mean1.unsqueeze(dim=2) - mean2.unsqueeze(dim=1)) * torch.reciprocal(var12)
            y = (-0.5 * (math.log(2.0 * math.pi) + torch.log(var12) + sq)).view(mean1.shape[0], -1)
        else:
            raise NotImplementedError(n.op)
        outs.append(y)
    return _select(outs, pg.output)


# --------------------------------------------------------------------------------------------
# layers -- cirkit/backend/torch/layers/{input,inner,optimized}.py
# --------------------------------------------------------------------------------------------
def _layer_forward(sr, l: LayerSpec, params: Mapping[str, Tensor], x) -> Tensor:
    t = l.type
    if t == "categorical":  # input.py:399-412 then :276-278 (map_from(LSE))
        xi = x.long() if x.is_floating_point() else x
        xi = xi.squeeze(dim=2)
        logits = torch.log(params["probs"]) if "probs" in params else params["logits"]
        idx_fold = torch.arange(l.num_folds)
        return sr.from_lse(logits[idx_fold[:, None], :, xi])
    if t == "gaussian":  # input.py:661-670
        mean = params["mean"].unsqueeze(dim=1)
        stddev = params["stddev"].unsqueeze(dim=1)
        lp = torch.distributions.Normal(loc=mean, scale=stddev).log_prob(x)
        if "log_partition" in params:
            lp = lp + params["log_partition"].unsqueeze(dim=1)
        return sr.from_lse(lp)
    if t == "binomial":  # input.py:530-541
        xi = x.long() if x.is_floating_point() else x
        n = int(l.config["total_count"])
        if "probs" in params:
            dist = torch.distributions.Binomial(n, probs=params["probs"].unsqueeze(dim=1))
        else:
            dist = torch.distributions.Binomial(n, logits=params["logits"].unsqueeze(dim=1))
        return sr.from_lse(dist.log_prob(xi))
    if t == "embedding":  # input.py:258-266
        xi = x.long() if x.is_floating_point() else x
        xi = xi.squeeze(dim=2)
        w = params["weight"]
        idx_fold = torch.arange(l.num_folds)
        return sr.from_sum_product(w[idx_fold[:, None], :, xi])
    if t == "constant":  # input.py:739-743 ; x is the batch size
        v = params["value"]
        v = v.unsqueeze(dim=1).expand(v.shape[0], int(x), v.shape[1])
        return sr.from_lse(v) if l.config.get("log_space") else sr.from_sum_product(v)
    if t == "hadamard":  # inner.py:126-127
        return sr.prod(x, dim=1)
    if t == "kronecker":  # inner.py:178-187
        y0 = x[:, 0]
        for i in range(1, x.shape[1]):
            y0 = torch.flatten(sr.mul(y0.unsqueeze(dim=-1), x[:, i].unsqueeze(dim=-2)), start_dim=-2)
        return y0
    if t == "sum":  # inner.py:266-273
        xf = x.permute(0, 2, 1, 3).flatten(start_dim=2)
        return _einsum(sr, "fbi,foi->fbo", inputs=(xf,), operands=(params["weight"],), dim=-1, keepdim=True)
    if t == "cpt":  # optimized.py:171-178
        xp = sr.prod(x, dim=1)
        return _einsum(sr, "fbi,foi->fbo", inputs=(xp,), operands=(params["weight"],), dim=-1, keepdim=True)
    if t == "tensordot":  # optimized.py:287-300
        w = params["weight"]
        kj = w.shape[2]
        kq = l.num_input_units // kj
        xs = x.squeeze(dim=1)
        xs = xs.view(xs.shape[0], xs.shape[1], kj, kq).permute(0, 1, 3, 2)
        y = _einsum(sr, "fbqj,fkj->fbqk", inputs=(xs,), operands=(w,), dim=-1, keepdim=True)
        return y.reshape(y.shape[0], y.shape[1], l.num_output_units)
    if t == "tucker":  # optimized.py:57-103: einsum (f,b,i_0), .., (f,b,i_{H-1}), (f,o,i_0..i_{H-1}) -> (f,b,o)
        H = l.arity
        w = params["weight"].view(-1, l.num_output_units, *(l.num_input_units for _ in range(H)))
        ops = tuple(sr.cast(o) for o in (w,))
        sub = tuple((0, 1, i + 2) for i in range(H)) + ((0, H + 2, *tuple(i + 2 for i in range(H))),)

        def func(*xs: Tensor) -> Tensor:
            args = [a for pair in zip((*xs, *ops), sub) for a in pair]
            return torch.einsum(*args, (0, 1, H + 2))

        return _apply_reduce(sr, func, *x.unbind(dim=1), dim=-1, keepdim=True)
    raise NotImplementedError(t)


def _integrate_input(sr, l: LayerSpec, params, output: Tensor, mask: Tensor) -> Tensor:
    """IntegrateQuery._layer_fn, queries.py:103-150: where the (B, D) mask marks the layer's variable,
    replace the output by ``layer.integrate()`` (TorchExpFamilyLayer.integrate input.py:280-282;
    log_partition_function :414-421 for Categorical, :672-679 for Gaussian)."""
    if mask.dim() == 1:
        mask = mask.unsqueeze(0)
    m = mask[:, torch.from_numpy(l.scope_idx)]  # (B|1, F, 1)
    m = m.permute(1, 0, 2)
    if l.type == "categorical":
        if "probs" in params:
            integ = torch.zeros((l.num_folds, 1, l.num_output_units), dtype=output.dtype)
        else:
            integ = torch.logsumexp(params["logits"], dim=2).unsqueeze(1)
    elif l.type == "gaussian":
        if "log_partition" in params:
            integ = params["log_partition"].unsqueeze(1)
        else:
            integ = torch.zeros((l.num_folds, 1, l.num_output_units), dtype=output.dtype)
    elif l.type == "binomial":  # input.py:543-549
        integ = torch.zeros((l.num_folds, 1, l.num_output_units), dtype=output.dtype)
    else:
        raise NotImplementedError(f"integrate() of a {l.type} layer")
    return torch.where(m, sr.from_lse(integ.to(output.real.dtype)), output)


def evaluate_plan(
    plan: Plan,
    tensors: Mapping[str, Tensor],
    x: Tensor | None,
    *,
    return_all: bool = False,
    grad: bool = False,
    integrate_mask: Tensor | None = None,
):
    """TorchCircuit.forward, circuits.py:242-278 + the interpreter loop graph/modules.py:303-335.

    Returns the circuit output ``(B, O, K)`` (``(O, K)`` for an empty-scope circuit); with
    ``return_all`` also the list of every layer's ``(F, B, Ko)`` output."""
    sr = _CLSE if plan.semiring == "complex-lse-sum" else _LSE
    outs: list[Tensor] = []
    # grad=True keeps the autograd graph: the reference trains by back-propagating through exactly
    # these ops (notebooks/learning-a-circuit.ipynb, `loss = -torch.mean(circuit(batch))`)
    with torch.enable_grad() if grad else torch.no_grad():
        for l in plan.layers:
            params = {pn: eval_param(pg, tensors) for pn, pg in l.params.items()}
            if l.inputs is not None:  # circuits.py:39-48
                xin = _select(outs, l.inputs)
            elif l.type == "constant":  # circuits.py:71
                xin = 1 if x is None else x.shape[0]
            else:  # circuits.py:66
                assert x is not None and x.dim() == 2
                xin = x[..., torch.from_numpy(l.scope_idx)].permute(1, 0, 2)
            y_l = _layer_forward(sr, l, params, xin)
            if integrate_mask is not None and l.inputs is None and l.type != "constant":
                y_l = _integrate_input(sr, l, params, y_l, integrate_mask)
            outs.append(y_l)
        y = _select(outs, plan.output)  # (O, B, K)
        y = y.transpose(0, 1)
        if plan.num_variables == 0:
            y = y.squeeze(dim=0)
    return (y, outs) if return_all else y


def as_torch(tensors: Mapping[str, np.ndarray]) -> dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in tensors.items()}
