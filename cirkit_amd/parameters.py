"""Device-side evaluation of layer parameter graphs.

Host mirror of ``cirkit.backend.torch.parameters`` for the nodes on the hot path (SURVEY.md
section 8 a13): the same node vocabulary (tensor, pointer, softmax, scaled_sigmoid, mixing_weight,
conj, matmul, einsum, flatten ...) evaluated with the ``ck_param_*`` HIP kernels on fold-stacked
blocks.  torch tensors are used as device storage only.

Reference: ``TorchParameter.forward`` cirkit/backend/torch/parameters/parameter.py:180-188, node
forwards in parameters/nodes.py (line numbers cited per op below).
"""

from __future__ import annotations

from typing import Mapping

import numpy as np
import torch

from . import _capi as capi
from .plan import IDX_ARRAY, IDX_NONE, FoldIndex, ParamGraph, resolve_fold_index


def _ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


class TensorStore:
    """Named parameter tensors on one device -- the counterpart of the ``nn.Parameter`` owned by
    ``TorchTensorParameter`` (nodes.py:188-201).  In-place updates are seen by the next forward;
    replacing a tensor object requires re-binding the circuits that use it (`version` bumps)."""

    def __init__(self, device: torch.device | str):
        self.device = torch.device(device)
        self._t: dict[str, torch.Tensor] = {}
        self.version = 0  # bumps when a tensor OBJECT is replaced (recorded pointers go stale)
        self.data_version = 0  # bumps on every value change (derived-parameter caches go stale)
        self._volatile = 0  # `state()` calls that could not read a version counter (inference tensors)
        # set by a HipCircuit that padded its unit counts (cirkit_amd/padding.py): values arrive and leave in
        # the shapes of the user's plan
        self._pad = None
        self._padded: dict[int, tuple] = {}  # id(user plan) -> (user plan, padded plan, PadInfo)

    def state(self) -> tuple:
        """What a cache of derived parameters compares: `data_version` (set / touch) and the sum of the tensors' torch
        version counters -- any in-place torch operation on a stored tensor (an optimizer step, `store[name].mul_(..)`)
        bumps one.  Writes through raw pointers by foreign kernels are invisible to both: call `touch()` after them."""
        total = 0
        for t in self._t.values():
            try:
                total += t._version
            except RuntimeError:
                # a tensor created under torch.inference_mode() has no version counter: its in-place updates cannot be
                # seen, so no state of this store ever equals an earlier one (derived parameters are re-evaluated at the
                # start of every forward, as the reference does)
                self._volatile += 1
                return (self.data_version, -self._volatile)
        return (self.data_version, total)

    def touch(self) -> None:
        """Record that tensor values were modified in place outside `set`."""
        self.data_version += 1

    def set(self, name: str, value) -> None:
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(np.ascontiguousarray(value))
        value = value.detach()
        if self._pad is not None and name in self._pad.shapes:
            old, new = self._pad.shapes[name]
            if old != new and tuple(value.shape) == tuple(old):
                value = torch.from_numpy(self._pad.pad(name, value.cpu().numpy()))
        self.data_version += 1
        if value.dtype not in (torch.float32, torch.complex64):
            value = value.to(torch.complex64 if value.is_complex() else torch.float32)
        cur = self._t.get(name)
        if cur is not None and cur.shape == value.shape and cur.dtype == value.dtype:
            cur.copy_(value)  # keep the pointer recorded programs hold
            return
        self._t[name] = value.to(self.device).contiguous()
        self.version += 1

    def update(self, values: Mapping[str, object]) -> None:
        for k, v in values.items():
            self.set(k, v)

    def export(self, name: str) -> np.ndarray:
        """The value of a tensor in the shape of the user's plan (host copy)."""
        v = self._t[name].detach().cpu().numpy()
        if self._pad is not None and name in self._pad.shapes and self._pad.shapes[name][0] != self._pad.shapes[name][1]:
            v = self._pad.unpad(name, v)
        return v

    def __getitem__(self, name: str) -> torch.Tensor:
        return self._t[name]

    def __contains__(self, name: str) -> bool:
        return name in self._t

    def names(self) -> list[str]:
        return list(self._t)


class ParamBatch:
    """All ``tensor -> softmax(last axis)`` parameters of a circuit, recomputed by ONE launch
    (`ck_param_softmax_batch`) at the start of every forward."""

    def __init__(self) -> None:
        self._jobs: list[tuple] = []
        self._keep: list[torch.Tensor] = []
        self._meta: list[dict] = []  # per job: its kind and tensors (for whoever takes a job over, HipCircuit._plan_tail_params)
        self._arr = None

    def subset(self, indices) -> "ParamBatch":
        """A batch of the listed jobs only (same tensors)."""
        pb = ParamBatch()
        for i in indices:
            pb._jobs.append(self._jobs[i])
            pb._meta.append(self._meta[i])
        pb._keep = list(self._keep)
        return pb

    def add_softmax(self, src: torch.Tensor, dst: torch.Tensor, layout: int = 0) -> None:
        """dst = softmax(src, dim=-1); both (..., len) contiguous fp32.  `layout` 1 writes the
        (F, 32, 32) result in the MFMA-tiled fp32 layout of ck_tile.h instead."""
        rows = src.numel() // src.shape[-1]
        kind = {0: 0, 1: 2}[layout]
        self._jobs.append((src.data_ptr(), dst.data_ptr(), rows, int(src.shape[-1]), 0, kind, None, None, None))
        self._meta.append({"kind": kind, "src": src, "dst": dst})
        self._keep += [src, dst]
        self._arr = None

    def add_log_table(self, src: torch.Tensor, dst: torch.Tensor) -> None:
        """src (F, K, C) logits -> dst (F, C+1, K) = log softmax over C, transposed; row C = 0."""
        F, K, Cc = src.shape
        self._jobs.append((src.data_ptr(), dst.data_ptr(), int(F), int(Cc), int(K), 1, None, None, None))
        self._meta.append({"kind": 1, "src": src, "dst": dst})
        self._keep += [src, dst]
        self._arr = None

    def add_log_table_dense(self, src: torch.Tensor, dense_src: torch.Tensor, idx: torch.Tensor | None,
                            dst: torch.Tensor, scale: torch.Tensor | None = None) -> None:
        """dst (Fd, C+1, 32) = the log-table of categorical fold idx[d] (src (F, 32, C) logits) pushed
        through dense fold d (dense_src (Fd, 32, 32) logits, softmax over the last axis).  With
        `scale` (Fd, C+1) the rows are left in linear space and their log scales go to `scale`."""
        F, K, Cc = src.shape
        if (K not in (32, 64) or tuple(dense_src.shape[1:]) != (K, K) or tuple(dst.shape) != (dense_src.shape[0], Cc + 1, K)
                or (K == 64 and scale is not None)):
            raise ValueError("add_log_table_dense needs K = 32 (or 64 without scales) and matching shapes")
        self._jobs.append((src.data_ptr(), dst.data_ptr(), int(dense_src.shape[0]), int(Cc), int(K), 4 if scale is None else 5,
                           dense_src.data_ptr(), None if idx is None else idx.data_ptr(),
                           None if scale is None else scale.data_ptr()))
        self._meta.append({"kind": 4 if scale is None else 5, "src": src, "dense": dense_src, "idx": idx, "dst": dst, "scale": scale})
        self._keep += [src, dense_src, dst] + ([] if idx is None else [idx]) + ([] if scale is None else [scale])
        self._arr = None

    def __len__(self) -> int:
        return len(self._jobs)

    def launch(self, stream: int) -> None:
        if not self._jobs:
            return
        if self._arr is None:
            self._arr = (capi.SoftmaxJob * len(self._jobs))()
            for a, (i, o, rows, ln, k, kind, in2, idx, out2) in zip(self._arr, self._jobs):
                a.inp, a.out, a.rows, a.len, a.k, a.kind, a.block_begin = i, o, rows, ln, k, kind, 0
                a.in2, a.idx, a.out2 = in2, idx, out2
        capi.call("ck_param_softmax_batch", self._arr, len(self._jobs), stream)


def _node_as_einsum(op: str, config: Mapping[str, Any], shapes: list[tuple[int, ...]]) -> tuple[tuple, tuple[int, ...]]:
    """Product-type parameter nodes as the einsum they are: (index tuples of the operands then of the output, the output's shape
    BEFORE the node's reshape).  hadamard (nodes.py:510-528): the same indices everywhere; kronecker (nodes.py:531-550, torch.kron
    per fold): out[(a_0, b_0), (a_1, b_1), ...] = x1[a...] x2[b...]; outer_product along `dim` (nodes.py:553-612):
    out[..., (i1, i2), ...] = x1[..., i1, ...] x2[..., i2, ...]; reduce_sum along `dim` (nodes.py:749-751)."""
    n = len(shapes[0])
    if op == "hadamard":
        idx = tuple(range(n))
        return (idx, idx, idx), tuple(shapes[0])
    if op == "kronecker":
        a, b = tuple(range(0, 2 * n, 2)), tuple(range(1, 2 * n, 2))
        out = tuple(range(2 * n))
        return (a, b, out), tuple(d for pair in zip(shapes[0], shapes[1]) for d in pair)
    d = int(config.get("dim", -1))
    d = d if d >= 0 else d + n
    if op == "outer_product":
        a = tuple(range(n))
        b = tuple(n if i == d else i for i in range(n))
        out = tuple(a[:d]) + (d, n) + tuple(a[d + 1:])
        return (a, b, out), tuple(shapes[0][:d]) + (shapes[0][d], shapes[1][d]) + tuple(shapes[0][d + 1:])
    if op == "reduce_sum":
        a = tuple(range(n))
        return (a, tuple(i for i in a if i != d)), tuple(s for i, s in enumerate(shapes[0]) if i != d)
    raise NotImplementedError(op)


_EINSUM_NODES = ("hadamard", "kronecker", "outer_product", "reduce_sum")


def _einsum_as_bmm(einsum, shapes):
    """Map a two-operand einsum over per-fold matrices onto ck_param_bmm; returns
    (swap, M, N, Kd, trans_a, trans_b) or None."""
    if len(einsum) != 3 or any(len(e) != 2 for e in einsum):
        return None
    a, b, o = (tuple(e) for e in einsum)
    for swap in (False, True):
        x, y = (b, a) if swap else (a, b)
        sx, sy = (shapes[1], shapes[0]) if swap else (shapes[0], shapes[1])
        m, n = o
        if m in x and n in y:
            kx = [i for i in x if i != m]
            ky = [i for i in y if i != n]
            if len(kx) == 1 and kx == ky and kx[0] not in o:
                trans_a = 0 if x.index(m) == 0 else 1
                trans_b = 0 if y.index(n) == 1 else 1
                M = sx[x.index(m)]
                N = sy[y.index(n)]
                Kd = sx[x.index(kx[0])]
                return swap, M, N, Kd, trans_a, trans_b
    return None


class HipParameter:
    """A parameter graph bound to a TensorStore.  ``evaluate()`` enqueues the kernels that
    recompute it (as the reference does on every forward) and returns the device tensor
    ``(F, *shape)``; buffers are allocated once so recorded programs can replay the launches."""

    def __init__(self, graph: ParamGraph, store: TensorStore):
        self.graph = graph
        self.store = store
        self._bufs: dict[object, torch.Tensor] = {}
        self._idx: dict[object, torch.Tensor] = {}

    def _gram_backward(self, n, xs, ins, out_idx, dj, sink, stream) -> bool:
        """The operand gradients of a Gram product ``y[f] = x[f] x[f]^T`` (both operands the SAME stored tensor: the Gram matrices
        of a squared circuit's input layers) as one accumulating launch, ``d x += (d y + d y^T) x`` -- autograd's two products
        ``d y x`` and ``d y^T x`` summed before the multiplication.  False: not that pattern (the caller takes the general path)."""
        if len(xs) != 2 or xs[0].data_ptr() != xs[1].data_ptr() or xs[0].shape != xs[1].shape:
            return False
        shapes = [tuple(dj.shape[1:]), tuple(xs[1].shape[1:])]
        m0 = _einsum_as_bmm([out_idx, ins[1], ins[0]], shapes)
        m1 = _einsum_as_bmm([out_idx, ins[0], ins[1]], shapes)
        if m0 is None or m1 is None or m0[0] or m1[0] or m0[1:4] != m1[1:4] or m0[5] != m1[5] or {m0[4], m1[4]} != {0, 1}:
            return False
        _, M, N, Kd, _, tb = m0
        if M != Kd or M % 32 or N % 32:
            return False
        F = int(xs[0].shape[0])
        d0, d1 = sink(n.inputs[0], F * M * N), sink(n.inputs[1], F * M * N)
        if d0 is None or d1 is None or d0.data_ptr() != d1.data_ptr():
            return False
        capi.call("ck_param_bmm", _ptr(dj.contiguous()), _ptr(xs[1].contiguous()), _ptr(d0), F, M, N, Kd, 2, tb, 1, stream)
        return True

    # -- surface mirrored from TorchParameter -------------------------------------------------
    @property
    def num_folds(self) -> int:
        return self.graph.num_folds

    @property
    def shape(self) -> tuple[int, ...]:
        return self.graph.shape

    @property
    def ops(self) -> list[str]:
        return self.graph.ops

    def __call__(self) -> torch.Tensor:
        return self.evaluate()

    # -- helpers ---------------------------------------------------------------------------------
    def _buf(self, key, shape, dtype=torch.float32) -> torch.Tensor:
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(tuple(shape), dtype=dtype, device=self.store.device)
            self._bufs[key] = t
        return t

    def _onehot(self, j: int, indices, n_in: int, F: int) -> torch.Tensor:
        """(F, len(indices), n_in) one-hot rows: TorchIndexParameter as a product the einsum launch (and its backward) takes."""
        key = ("onehot", j)
        t = self._idx.get(key)
        if t is None:
            m = np.zeros((F, len(indices), n_in), dtype=np.float32)
            m[:, np.arange(len(indices)), np.asarray(indices, dtype=np.int64)] = 1.0
            t = torch.from_numpy(m).to(self.store.device)
            self._idx[key] = t
        return t

    def _index_tensor(self, key, arr: np.ndarray) -> torch.Tensor:
        t = self._idx.get(key)
        if t is None:
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)).to(self.store.device)
            self._idx[key] = t
        return t

    def _gather(self, key, src: torch.Tensor, idx: np.ndarray, stream: int) -> torch.Tensor:
        per_fold = int(np.prod(src.shape[1:])) * (2 if src.is_complex() else 1)
        out = self._buf(("g", key), (len(idx), *src.shape[1:]), src.dtype)
        capi.call(
            "ck_param_gather_folds",
            _ptr(src),
            _ptr(self._index_tensor(("gi", key), idx)),
            _ptr(out),
            len(idx),
            per_fold,
            stream,
        )
        return out

    def _select(self, key, outs: list[torch.Tensor], fi: FoldIndex, stream: int) -> torch.Tensor:
        """parameter.py:41-47: cat the producers along folds, then index."""
        if len(fi.ids) == 1:
            src = outs[fi.ids[0]]
            if fi.kind == IDX_NONE:
                return src
            assert fi.kind == IDX_ARRAY
            idx = np.asarray(fi.array, dtype=np.int64).reshape(-1)
            if np.array_equal(idx, np.arange(src.shape[0])):
                return src
            return self._gather(key, src, idx, stream)
        # several producers: gather each producer's rows into the concatenation, then index it
        folds = {i: outs[i].shape[0] for i in fi.ids}
        pairs = resolve_fold_index(fi, [folds.get(i, 0) for i in range(max(fi.ids) + 1)])
        pairs = pairs.reshape(-1, 2)
        first = outs[fi.ids[0]]
        cat = self._buf(("cat", key), (sum(folds.values()), *first.shape[1:]), first.dtype)
        off = 0
        for i in fi.ids:
            n = folds[i]
            per_fold = int(np.prod(first.shape[1:])) * (2 if first.is_complex() else 1)
            capi.call(
                "ck_param_gather_folds",
                _ptr(outs[i]),
                _ptr(self._index_tensor(("ci", key, i), np.arange(n))),
                _ptr(cat[off : off + n]),
                n,
                per_fold,
                stream,
            )
            off += n
        if fi.kind == IDX_NONE:
            return cat
        return self._gather(key, cat, np.asarray(fi.array, dtype=np.int64).reshape(-1), stream)

    # -- evaluation ----------------------------------------------------------------------------
    def evaluate(self, stream: int = 0, *, upto: int | None = None) -> torch.Tensor:
        """Enqueue the graph on `stream`.  With ``upto=j`` return the output of node j instead of
        the graph output (used by layers that fuse the tail of the graph into their own kernel)."""
        g = self.graph
        outs: list[torch.Tensor] = []
        last = len(g.nodes) - 1 if upto is None else upto
        for j, n in enumerate(g.nodes[: last + 1]):
            xs = [self._select((j, k), outs, fi, stream) for k, fi in enumerate(n.inputs)]
            c = n.config
            shape = (n.num_folds, *n.shape)
            if n.op == "tensor":  # nodes.py:203-220
                y = self.store[c["tensor"]]
            elif n.op == "pointer":  # nodes.py:277-279
                y = self.store[c["tensor"]]
                if c.get("fold_idx") is not None:
                    y = self._gather(("p", j), y, np.asarray(c["fold_idx"], dtype=np.int64), stream)
            elif n.op in ("softmax", "log_softmax"):  # nodes.py:764-783
                x = xs[0]
                if x.is_complex():
                    raise NotImplementedError("softmax of a complex parameter")
                dim = int(c["dim"]) + 1
                outer = int(np.prod(x.shape[:dim]))
                inner = int(np.prod(x.shape[dim + 1 :]))
                y = self._buf(j, shape)
                capi.call(
                    "ck_param_softmax", _ptr(x), _ptr(y), outer, int(x.shape[dim]), inner,
                    1 if n.op == "log_softmax" else 0, stream,
                )
            elif n.op in ("sigmoid", "scaled_sigmoid", "exp", "log", "square", "clamp", "softplus"):  # nodes.py:656-739
                x = xs[0]
                if x.is_complex():
                    raise NotImplementedError(f"{n.op} of a complex parameter")
                code = {
                    "sigmoid": capi.CK_UNARY_SIGMOID,
                    "scaled_sigmoid": capi.CK_UNARY_SCALED_SIGMOID,
                    "exp": capi.CK_UNARY_EXP,
                    "log": capi.CK_UNARY_LOG,
                    "square": capi.CK_UNARY_SQUARE,
                    "clamp": capi.CK_UNARY_CLAMP,
                    "softplus": capi.CK_UNARY_SOFTPLUS,
                }[n.op]
                y = self._buf(j, shape)
                if n.op == "clamp":  # (an absent bound: no clamping on that side, nodes.py:727-728)
                    lo = float("-inf") if c.get("vmin") is None else float(c["vmin"])
                    hi = float("inf") if c.get("vmax") is None else float(c["vmax"])
                else:
                    lo, hi = float(c.get("vmin", 0.0)), float(c.get("vmax", 1.0))
                capi.call("ck_param_unary", code, _ptr(x), _ptr(y), x.numel(), lo, hi, stream)
            elif n.op == "conj":  # nodes.py:745-746
                x = xs[0]
                if x.is_complex():
                    y = self._buf(j, shape, torch.complex64)
                    capi.call("ck_param_conj", _ptr(x), _ptr(y), x.numel(), stream)
                else:
                    y = x  # conj of a real tensor is the tensor
            elif n.op == "mixing_weight":  # nodes.py:857-862
                x = xs[0]
                F, K, H = x.shape
                y = self._buf(j, (F, K, H * K))
                capi.call("ck_param_mixing_weight", _ptr(x), _ptr(y), F, K, H, stream)
            elif n.op == "matmul":  # nodes.py:802-805
                a, b = xs
                if a.is_complex() or b.is_complex():  # (the generic einsum: "ik,kj->ij" over complex pairs)
                    y = self._einsum(j, ((0, 1), (1, 2), (0, 2)), [a, b], stream)
                else:
                    F, M, Kd = a.shape
                    N = b.shape[2]
                    y = self._buf(j, (F, M, N))
                    capi.call("ck_param_bmm", _ptr(a), _ptr(b), _ptr(y), F, M, N, Kd, 0, 0, 0, stream)
            elif n.op == "einsum":  # optimized.py:282-284
                m = None
                if len(xs) == 2 and not any(x.is_complex() for x in xs):
                    m = _einsum_as_bmm(c["einsum"], [tuple(x.shape[1:]) for x in xs])
                if m is None:  # any other pattern, complex operands: the generic kernel
                    y = self._einsum(j, c["einsum"], xs, stream)
                else:
                    swap, M, N, Kd, ta, tb = m
                    a, b = (xs[1], xs[0]) if swap else (xs[0], xs[1])
                    y = self._buf(j, (a.shape[0], M, N))
                    capi.call("ck_param_bmm", _ptr(a), _ptr(b), _ptr(y), a.shape[0], M, N, Kd, ta, tb, 0, stream)
            elif n.op == "flatten":  # nodes.py:843-844
                y = xs[0].reshape(shape)
            elif n.op in _EINSUM_NODES:  # nodes.py:510-612, 749-751: products and sums over indices -- the generic einsum launch
                spec, _ = _node_as_einsum(n.op, c, [tuple(x.shape[1:]) for x in xs])
                y = self._einsum(j, spec, [x.contiguous() for x in xs], stream).reshape(shape)
            elif n.op in ("reduce_prod", "reduce_lse"):  # nodes.py:754-761
                x = xs[0].contiguous()
                if x.is_complex():
                    raise NotImplementedError(f"{n.op} of a complex parameter")
                d = int(c.get("dim", -1))
                d = (d if d >= 0 else d + x.dim() - 1) + 1
                y = self._buf(j, shape)
                capi.call("ck_param_reduce", 0 if n.op == "reduce_prod" else 1, _ptr(x), _ptr(y), int(np.prod(x.shape[:d])), int(x.shape[d]),
                          int(np.prod(x.shape[d + 1:])), stream)
            elif n.op == "outer_sum":  # nodes.py:615-653
                a, b = (x.contiguous() for x in xs)
                if a.is_complex() or b.is_complex():
                    raise NotImplementedError("outer sum of complex parameters")
                d = int(c.get("dim", -1))
                d = (d if d >= 0 else d + a.dim() - 1) + 1
                y = self._buf(j, shape)
                capi.call("ck_param_outer_sum", _ptr(a), _ptr(b), _ptr(y), int(np.prod(a.shape[:d])), int(a.shape[d]), int(b.shape[d]),
                          int(np.prod(a.shape[d + 1:])), stream)
            elif n.op == "index":  # nodes.py:450-488: x[:, indices] -- the first axis of the per-fold value, as the reference's forward does
                x = xs[0].contiguous()
                if x.is_complex():
                    raise NotImplementedError("index of a complex parameter")
                sel = self._onehot(j, c["indices"], int(x.shape[1]), int(x.shape[0]))
                rest = tuple(range(2, x.dim()))
                spec = ((0, 1), (1, *rest), (0, *rest))
                y = self._einsum(j, spec, [sel, x], stream)
            elif n.op == "sum":  # nodes.py:491-507: x1 + x2
                a, b = xs
                if a.is_complex() or b.is_complex():
                    raise NotImplementedError("sum of complex parameters")
                y = self._buf(j, shape)
                capi.call("ck_copy_strided_f32", _ptr(a.contiguous()), _ptr(y), y.numel(), 1, 1, stream)
                capi.call("ck_axpy_f32", _ptr(y), _ptr(b.contiguous()), 1.0, y.numel(), stream)
            elif n.op in ("gaussian_product_mean", "gaussian_product_stddev"):  # nodes.py:865-938
                ops = [x.contiguous() for x in xs]
                if any(x.dim() > 2 and int(np.prod(x.shape[2:])) != 1 for x in ops):
                    raise NotImplementedError(f"{n.op} over several channels")
                mean = n.op == "gaussian_product_mean"
                m1, s1, m2, s2 = ops if mean else (None, ops[0], None, ops[1])
                F, K1, K2 = int(s1.shape[0]), int(s1.shape[1]), int(s2.shape[1])
                y = self._buf(j, shape)  # ((F, K1 K2): `shape` holds the fold axis already)
                capi.call("ck_param_gaussian_product_ms", 0 if mean else 1, None if m1 is None else _ptr(m1), _ptr(s1),
                          None if m2 is None else _ptr(m2), _ptr(s2), _ptr(y), F, K1, K2, stream)
            elif n.op == "gaussian_product_log_partition":  # nodes.py:975-988
                m1, s1, m2, s2 = (x.contiguous() for x in xs)
                F, K1, K2 = m1.shape[0], m1.shape[1], m2.shape[1]
                y = self._buf(j, (F, K1 * K2))
                capi.call("ck_param_gaussian_product_logz", _ptr(m1), _ptr(s1), _ptr(m2), _ptr(s2), _ptr(y), F, K1, K2, stream)
            else:
                raise NotImplementedError(f"parameter op {n.op}")
            outs.append(y)
        self._last_outs = outs  # (what `backward` differentiates: nodes that alias their input own no buffer of their own)
        if upto is not None:
            return outs[upto]
        return self._select("out", outs, g.output, stream)

    def _einsum(self, j: int, einsum, xs: list[torch.Tensor], stream: int) -> torch.Tensor:
        """`ck_param_einsum`: TorchEinsumParameter.forward (parameters/optimized.py:282-284) for any index pattern over up to
        four real or complex operands (F, *shape); `einsum` = the operands' index tuples, then the output's."""
        import ctypes as C

        *ins, out_idx = [tuple(int(i) for i in e) for e in einsum]
        if len(ins) != len(xs) or not 1 <= len(xs) <= 4:
            raise NotImplementedError(f"einsum parameter over {len(xs)} operands")
        extent: dict[int, int] = {}
        for e, x in zip(ins, xs):
            if len(e) != x.dim() - 1:
                raise ValueError(f"einsum indices {e} against an operand of shape {tuple(x.shape)}")
            for lab, n in zip(e, x.shape[1:]):
                if extent.setdefault(lab, int(n)) != int(n):
                    raise ValueError(f"einsum index {lab}: extents {extent[lab]} and {int(n)}")
        if len(set(out_idx)) != len(out_idx) or any(lab not in extent for lab in out_idx):
            raise ValueError(f"einsum output indices {out_idx}")
        order = list(out_idx) + sorted(lab for lab in extent if lab not in out_idx)
        if len(order) > 8:
            raise NotImplementedError(f"einsum parameter over {len(order)} indices")
        F = int(xs[0].shape[0])
        cplx = any(x.is_complex() for x in xs)
        if not all(x.is_contiguous() for x in xs):  # (a torch copy here would not be part of the recorded launches)
            raise NotImplementedError("einsum parameter over a non-contiguous operand")
        y = self._buf(j, (F, *(extent[lab] for lab in out_idx)), torch.complex64 if cplx else torch.float32)
        d = capi.EinsumDesc()
        d.out, d.n_ops, d.n_idx, d.n_out, d.F, d.out_complex = _ptr(y), len(xs), len(order), len(out_idx), F, 1 if cplx else 0
        for i, lab in enumerate(order):
            d.extent[i] = extent[lab]
        for k, (e, x) in enumerate(zip(ins, xs)):
            if int(x.shape[0]) != F:
                raise ValueError("einsum operands with different numbers of folds")
            d.x[k], d.is_complex[k] = _ptr(x), 1 if x.is_complex() else 0
            d.fold_stride[k] = int(np.prod(x.shape[1:]))
            strides = [int(np.prod(x.shape[2 + a:])) for a in range(len(e))]
            for a, lab in enumerate(e):
                d.stride[k][order.index(lab)] += strides[a]  # (a repeated index walks the diagonal)
        capi.call("ck_param_einsum", C.byref(d), stream)
        return y

    # -- backward -----------------------------------------------------------------------------
    def backward(self, dout: torch.Tensor, grads: Mapping[str, torch.Tensor], stream: int = 0, *,
                 upto: int | None = None) -> None:
        """Reverse-mode pass over the graph: `dout` is the gradient w.r.t. the value `evaluate`
        returned last (with the same `upto`); the gradients of the parameter tensors are ADDED into
        `grads[name]`.  Covers what the region-graph templates build -- tensor, softmax (last axis),
        scaled sigmoid, mixing weight, matmul -- including the fold re-indexing between nodes that
        folding introduces when the layers of a group carry different parameter graphs."""
        g = self.graph
        last = len(g.nodes) - 1 if upto is None else upto
        folds = [n.num_folds for n in g.nodes]

        def value(j: int) -> torch.Tensor:
            n = g.nodes[j]
            if n.op == "tensor":
                return self.store[n.config["tensor"]]
            outs = getattr(self, "_last_outs", None)
            return outs[j] if outs is not None and j < len(outs) else self._bufs[j]

        gb: dict[int, torch.Tensor] = {}

        def target(i: int) -> torch.Tensor:
            """The (accumulating) gradient buffer of node i: the caller's for tensors, a zeroed scratch otherwise."""
            n = g.nodes[i]
            if n.op == "tensor":
                return grads[n.config["tensor"]]
            if i not in gb:
                gb[i] = self._buf(("grad", i), value(i).shape)
                capi.call("ck_fill_f32", _ptr(gb[i]), gb[i].numel(), 0.0, stream)
            return gb[i]

        def is_identity(fi: FoldIndex) -> bool:
            return len(fi.ids) == 1 and (fi.kind == IDX_NONE or (
                fi.kind == IDX_ARRAY and np.array_equal(np.asarray(fi.array).reshape(-1), np.arange(folds[fi.ids[0]]))))

        # how many places read node i (operands of later nodes, the output): a node read ONCE, through an identity index, takes
        # its reader's gradient buffer as its own -- no zeroed scratch, no add (conj of a real value, flatten, the pointer chains
        # of a squared circuit's partition function: one launch instead of five per parameter)
        readers = [0] * len(g.nodes)
        for j, n in enumerate(g.nodes[: last + 1]):
            for fi in n.inputs:
                for i in fi.ids:
                    readers[i] += 1
        if upto is None:
            for i in g.output.ids:
                readers[i] += 1

        def scatter(key, fi: FoldIndex, dgathered: torch.Tensor) -> None:
            """Backward of `_select`: add the rows of `dgathered` into the producers they were gathered from."""
            if is_identity(fi):
                i = fi.ids[0]
                if g.nodes[i].op != "tensor" and readers[i] == 1 and i not in gb and dgathered.is_contiguous():
                    gb[i] = dgathered.view(value(i).shape) if dgathered.numel() == value(i).numel() else dgathered
                    return
                t = target(fi.ids[0])
                if t.data_ptr() != dgathered.data_ptr():
                    capi.call("ck_axpy_f32", _ptr(t), _ptr(dgathered), 1.0, dgathered.numel(), stream)
                return
            pairs = resolve_fold_index(fi, [folds[i] if i in fi.ids else 0 for i in range(max(fi.ids) + 1)]).reshape(-1, 2)
            per_fold = dgathered.numel() // dgathered.shape[0]
            for p in fi.ids:
                rows = np.nonzero(pairs[:, 0] == p)[0]
                if len(rows) == 0:
                    continue
                if np.array_equal(rows, np.arange(rows[0], rows[0] + len(rows))):
                    src = dgathered[rows[0] : rows[0] + len(rows)]
                else:
                    src = self._gather(("gs", key, p), dgathered, rows, stream)
                capi.call("ck_param_scatter_add_folds", _ptr(src), _ptr(self._index_tensor(("si", key, p), pairs[rows, 1])),
                          _ptr(target(p)), len(rows), per_fold, stream)

        def operand(j: int, k: int) -> torch.Tensor:
            fi = g.nodes[j].inputs[k]
            return value(fi.ids[0]) if is_identity(fi) else self._bufs[("g", (j, k))]

        def sink(fi: FoldIndex, numel: int):
            """The stored tensor's gradient that a REAL gradient handed to `fi` ends up added to, element for element -- through
            identity indices and conj (of real values) / flatten nodes down to a tensor node or a whole-tensor pointer (every
            one of them the identity map on the entries, so whatever else those nodes collect travels separately) -- or None:
            the producer then adds into it directly (`ck_param_bmm` with `accumulate`) instead of leaving a buffer that travels down the
            chain through zero-filled node gradients and axpys."""
            while is_identity(fi):
                i = fi.ids[0]
                nd = g.nodes[i]
                if nd.op == "tensor" or (nd.op == "pointer" and nd.config.get("fold_idx") is None):
                    t = grads[nd.config["tensor"]]
                elif nd.op in ("conj", "flatten") and not value(i).is_complex():
                    fi = nd.inputs[0]
                    continue
                else:
                    return None
                return t if (not t.is_complex() and t.is_contiguous() and t.numel() == numel) else None
            return None

        def direct(j: int, k: int):
            """(buffer, accumulate flag, needs scatter) for the gradient of operand k of node j."""
            fi = g.nodes[j].inputs[k]
            if is_identity(fi):
                i = fi.ids[0]
                if g.nodes[i].op != "tensor" and readers[i] == 1 and i not in gb:  # its only gradient: written, not added
                    gb[i] = self._buf(("grad", i), value(i).shape)
                    return gb[i], 0, False
                return target(i), 1, False
            return self._buf(("gop", j, k), operand(j, k).shape), 0, True

        if upto is None:
            scatter("out", g.output, dout)
        else:
            gb[upto] = dout
        for j in range(last, -1, -1):
            n = g.nodes[j]
            if n.op == "tensor" or j not in gb:
                continue
            dj = gb[j]
            if n.op in ("softmax", "log_softmax"):  # any axis (nodes.py:764-783)
                y = value(j)
                dx, acc, sc = direct(j, 0)
                dim = int(n.config["dim"]) + 1
                if n.op == "softmax" and dim == y.dim() - 1:
                    capi.call("ck_param_softmax_bwd", _ptr(y), _ptr(dj), _ptr(dx), y.numel() // y.shape[-1], int(y.shape[-1]), acc, stream)
                else:
                    capi.call("ck_param_softmax_bwd_strided", _ptr(y), _ptr(dj), _ptr(dx), int(np.prod(y.shape[:dim])), int(y.shape[dim]),
                              int(np.prod(y.shape[dim + 1:])), 1 if n.op == "log_softmax" else 0, acc, stream)
                if sc:
                    scatter((j, 0), n.inputs[0], dx)
            elif n.op in ("sigmoid", "exp", "log", "square", "clamp", "softplus"):  # entrywise nodes (nodes.py:656-739)
                y, x = value(j), operand(j, 0)
                dx, acc, sc = direct(j, 0)
                code = {"sigmoid": capi.CK_UNARY_SIGMOID, "exp": capi.CK_UNARY_EXP, "log": capi.CK_UNARY_LOG, "square": capi.CK_UNARY_SQUARE,
                        "clamp": capi.CK_UNARY_CLAMP, "softplus": capi.CK_UNARY_SOFTPLUS}[n.op]
                capi.call("ck_param_unary_bwd", code, _ptr(x), _ptr(y), _ptr(dj), _ptr(dx), y.numel(), acc, stream)
                if sc:
                    scatter((j, 0), n.inputs[0], dx)
            elif n.op == "scaled_sigmoid":
                y = value(j)
                dx, acc, sc = direct(j, 0)
                capi.call("ck_param_scaled_sigmoid_bwd", _ptr(y), _ptr(dj), _ptr(dx), y.numel(),
                          float(n.config.get("vmin", 0.0)), float(n.config.get("vmax", 1.0)), acc, stream)
                if sc:
                    scatter((j, 0), n.inputs[0], dx)
            elif n.op == "mixing_weight":
                F, K, H = operand(j, 0).shape
                dx, acc, sc = direct(j, 0)
                capi.call("ck_param_mixing_weight_bwd", _ptr(dj), _ptr(dx), F, K, H, acc, stream)
                if sc:
                    scatter((j, 0), n.inputs[0], dx)
            elif n.op == "matmul":  # y = a b: da = dy b^T, db = a^T dy
                a, b = operand(j, 0), operand(j, 1)
                F, M, Kd = a.shape
                N = b.shape[2]
                da = self._buf(("gtmp", j, 0), (F, M, Kd))
                db = self._buf(("gtmp", j, 1), (F, Kd, N))
                capi.call("ck_param_bmm", _ptr(dj), _ptr(b), _ptr(da), F, M, Kd, N, 0, 1, 0, stream)
                capi.call("ck_param_bmm", _ptr(a), _ptr(dj), _ptr(db), F, Kd, N, M, 1, 0, 0, stream)
                scatter((j, 0), n.inputs[0], da)
                scatter((j, 1), n.inputs[1], db)
            elif n.op == "pointer":  # a gather of folds of a stored tensor (nodes.py:277-279): scatter-add back
                t = grads[n.config["tensor"]]
                if dj.is_complex() or t.is_complex():
                    raise NotImplementedError("parameter backward through a complex pointer")
                idx = n.config.get("fold_idx")
                if idx is None:
                    capi.call("ck_axpy_f32", _ptr(t), _ptr(dj), 1.0, dj.numel(), stream)
                else:
                    capi.call("ck_param_scatter_add_folds", _ptr(dj), _ptr(self._index_tensor(("pi", j), np.asarray(idx, dtype=np.int64))),
                              _ptr(t), len(idx), dj.numel() // dj.shape[0], stream)
            elif n.op == "conj":  # (of a real value: the identity, nodes.py:745-746)
                if dj.is_complex() or operand(j, 0).is_complex():
                    raise NotImplementedError("parameter backward through the conjugate of a complex value")
                scatter((j, 0), n.inputs[0], dj)
            elif n.op == "flatten":  # nodes.py:843-844
                scatter((j, 0), n.inputs[0], dj.reshape(operand(j, 0).shape))
            elif n.op == "sum":  # nodes.py:491-507
                scatter((j, 0), n.inputs[0], dj)
                scatter((j, 1), n.inputs[1], dj)
            elif n.op in ("reduce_prod", "reduce_lse"):
                x, y = operand(j, 0).contiguous(), value(j)
                d = int(n.config.get("dim", -1))
                d = (d if d >= 0 else d + x.dim() - 1) + 1
                dx = self._buf(("gtmp", j, 0), x.shape)
                capi.call("ck_param_reduce_bwd", 0 if n.op == "reduce_prod" else 1, _ptr(x), _ptr(y), _ptr(dj.contiguous()), _ptr(dx),
                          int(np.prod(x.shape[:d])), int(x.shape[d]), int(np.prod(x.shape[d + 1:])), stream)
                scatter((j, 0), n.inputs[0], dx)
            elif n.op == "outer_sum":
                a, b = operand(j, 0), operand(j, 1)
                d = int(n.config.get("dim", -1))
                d = (d if d >= 0 else d + a.dim() - 1) + 1
                outer, inner = int(np.prod(a.shape[:d])), int(np.prod(a.shape[d + 1:]))
                for k, x in enumerate((a, b)):
                    dx = self._buf(("gtmp", j, k), x.shape)
                    capi.call("ck_param_outer_sum_bwd", _ptr(dj.contiguous()), _ptr(dx), outer, int(a.shape[d]), int(b.shape[d]), inner, k, stream)
                    scatter((j, k), n.inputs[k], dx)
            elif n.op == "index":  # y = S x with S the one-hot selection: dx = S^T dy
                x = operand(j, 0)
                sel = self._onehot(j, n.config["indices"], int(x.shape[1]), int(x.shape[0]))
                rest = tuple(range(2, x.dim()))
                dx = self._einsum(("ge", j, 0), ((0, 1), (0, *rest), (1, *rest)), [sel, dj.contiguous()], stream)
                scatter((j, 0), n.inputs[0], dx)
            elif n.op == "reduce_sum":  # nodes.py:749-751: the gradient is the output's, repeated along the summed axis
                x = operand(j, 0)
                if dj.is_complex() or x.is_complex():
                    raise NotImplementedError("parameter backward through a sum over complex entries")
                (a_idx, o_idx), _ = _node_as_einsum("reduce_sum", n.config, [tuple(x.shape[1:])])
                d = next(i for i in a_idx if i not in o_idx)
                ones = self._buf(("ones", j), (x.shape[0], x.shape[1 + d]))
                capi.call("ck_fill_f32", _ptr(ones), ones.numel(), 1.0, stream)
                dx = self._einsum(("ge", j, 0), (o_idx, (d,), a_idx), [dj.contiguous(), ones], stream)
                scatter((j, 0), n.inputs[0], dx)
            elif n.op == "einsum" or n.op in _EINSUM_NODES:  # y = sum over the contracted indices of prod_k x_k: d x_k = the same sum with y's gradient in x_k's place
                xs = [operand(j, k) for k in range(len(n.inputs))]
                if dj.is_complex() or any(x.is_complex() for x in xs):
                    raise NotImplementedError("parameter backward through an einsum over complex operands")
                if n.op == "einsum":
                    spec_all = n.config["einsum"]
                else:  # (the node's output is a reshape of the einsum's: its gradient back in the einsum's shape)
                    spec_all, eshape = _node_as_einsum(n.op, n.config, [tuple(x.shape[1:]) for x in xs])
                    dj = dj.contiguous().reshape(dj.shape[0], *eshape)
                *ins, out_idx = [tuple(int(i) for i in e) for e in spec_all]
                if self._gram_backward(n, xs, ins, out_idx, dj, sink, stream):  # (y = x x^T: d x = (d y + d y^T) x, one launch)
                    continue
                for k in range(len(xs)):
                    others = [m for m in range(len(xs)) if m != k]
                    have = set(out_idx) | {i for m in others for i in ins[m]}
                    if not set(ins[k]) <= have or len(set(ins[k])) != len(ins[k]):
                        raise NotImplementedError(f"parameter backward through einsum {spec_all}: operand {k} carries an index "
                                                  "that is summed out on its own or repeated")
                    spec = [out_idx, *[ins[m] for m in others], ins[k]]
                    ops_k = [dj.contiguous(), *[xs[m] for m in others]]
                    mb = _einsum_as_bmm(spec, [tuple(x.shape[1:]) for x in ops_k]) if len(ops_k) == 2 else None
                    if mb is None:
                        dk = self._einsum(("ge", j, k), spec, ops_k, stream)
                    else:  # a product of per-fold matrices (the Gram / Kronecker parameters of a squared circuit): the MFMA-free
                        # batched matmul instead of the index-generic kernel (0.85 ms -> tens of us per node at config 5)
                        swap, M, N, Kd, ta, tb = mb
                        a, b = (ops_k[1], ops_k[0]) if swap else (ops_k[0], ops_k[1])
                        a, b = a.contiguous(), b.contiguous()
                        dst = sink(n.inputs[k], a.shape[0] * M * N)
                        if dst is not None:  # (the Gram parameters of a squared circuit: 2 x (25 MB written + an axpy over 75 MB) less per step)
                            capi.call("ck_param_bmm", _ptr(a), _ptr(b), _ptr(dst), a.shape[0], M, N, Kd, ta, tb, 1, stream)
                            continue
                        dk = self._buf(("ge", j, k), (a.shape[0], M, N))
                        capi.call("ck_param_bmm", _ptr(a), _ptr(b), _ptr(dk), a.shape[0], M, N, Kd, ta, tb, 0, stream)
                    scatter((j, k), n.inputs[k], dk)
            elif n.op in ("gaussian_product_mean", "gaussian_product_stddev"):  # nodes.py:865-938
                mean = n.op == "gaussian_product_mean"
                ops = [operand(j, k).contiguous() for k in range(len(n.inputs))]
                m1, s1, m2, s2 = ops if mean else (None, ops[0], None, ops[1])
                F, K1, K2 = int(s1.shape[0]), int(s1.shape[1]), int(s2.shape[1])
                ds = [self._buf(("gop", j, k), ops[k].shape) for k in range(len(ops))]
                dm1, ds1, dm2, ds2 = ds if mean else (None, ds[0], None, ds[1])
                capi.call("ck_param_gaussian_product_ms_bwd", 0 if mean else 1, None if m1 is None else _ptr(m1), _ptr(s1),
                          None if m2 is None else _ptr(m2), _ptr(s2), _ptr(dj.contiguous()), None if dm1 is None else _ptr(dm1), _ptr(ds1),
                          None if dm2 is None else _ptr(dm2), _ptr(ds2), F, K1, K2, stream)
                for k in range(len(ops)):
                    scatter((j, k), n.inputs[k], ds[k])
            elif n.op == "gaussian_product_log_partition":  # nodes.py:975-988
                m1, s1, m2, s2 = (operand(j, k).contiguous() for k in range(4))
                F, K1, K2 = int(m1.shape[0]), int(m1.shape[1]), int(m2.shape[1])
                ds = [self._buf(("gop", j, k), operand(j, k).shape) for k in range(4)]
                capi.call("ck_param_gaussian_product_logz_bwd", _ptr(m1), _ptr(s1), _ptr(m2), _ptr(s2), _ptr(dj.contiguous()),
                          _ptr(ds[0]), _ptr(ds[1]), _ptr(ds[2]), _ptr(ds[3]), F, K1, K2, stream)
                for k in range(4):
                    scatter((j, k), n.inputs[k], ds[k])
            else:
                raise NotImplementedError(f"parameter backward through {n.op!r}")

    def passthrough_tensor(self) -> str | None:
        """The name of the stored REAL tensor that `evaluate` returns as it is -- a tensor or a pointer to all its folds, seen
        through conjugates and identity fold indices (the weights of a squared circuit's partition function) -- else None.
        A layer whose weight passes through may add its weight gradient straight into that tensor's gradient."""
        g = self.graph
        if not g.nodes or g.nodes[0].op not in ("tensor", "pointer") or g.nodes[0].config.get("fold_idx") is not None:
            return None
        for j, n in enumerate(g.nodes[1:], start=1):
            if n.op != "conj" or len(n.inputs) != 1:
                return None
            fi = n.inputs[0]
            if fi.ids != [j - 1] or not (fi.kind == IDX_NONE or (
                    fi.kind == IDX_ARRAY and np.array_equal(np.asarray(fi.array).reshape(-1), np.arange(g.nodes[j - 1].num_folds)))):
                return None
        if g.output.ids != [len(g.nodes) - 1] or not (g.output.kind == IDX_NONE or np.array_equal(
                np.asarray(g.output.array).reshape(-1), np.arange(g.nodes[-1].num_folds))):
            return None
        name = g.nodes[0].config["tensor"]
        return None if self.store[name].is_complex() else name

    def softmax_source(self) -> torch.Tensor | None:
        """The raw tensor when the graph is exactly ``tensor -> softmax(last axis)`` with identity
        fold indices (the default parameterisation of sum weights and Categorical probs), else None."""
        g = self.graph
        if g.ops != ["tensor", "softmax"] or not self.tail_is("softmax"):
            return None
        n = g.nodes[1]
        if int(n.config["dim"]) != len(n.shape) - 1:
            return None
        fi = n.inputs[0]
        if fi.ids != [0] or fi.kind != IDX_NONE:
            return None
        t = self.store[g.nodes[0].config["tensor"]]
        return None if t.is_complex() else t

    def mixing_softmax_source(self) -> torch.Tensor | None:
        """The raw (F, K, H) tensor when the graph is exactly ``tensor -> softmax(last axis) -> mixing_weight`` with
        identity fold indices (the default parameterisation of a mixing layer), else None."""
        g = self.graph
        if g.ops != ["tensor", "softmax", "mixing_weight"] or not self.tail_is("mixing_weight"):
            return None
        n = g.nodes[1]
        if int(n.config["dim"]) != len(n.shape) - 1:
            return None
        for node, src in ((g.nodes[1], 0), (g.nodes[2], 1)):
            fi = node.inputs[0]
            if fi.ids != [src] or fi.kind != IDX_NONE:
                return None
        t = self.store[g.nodes[0].config["tensor"]]
        return None if t.is_complex() else t

    def tail_is(self, *ops: str) -> bool:
        """True when the graph ends with `ops` feeding the output untouched (identity output index
        over a single producer) -- lets a layer fuse the tail into its kernel."""
        g = self.graph
        if len(g.nodes) < len(ops) or [n.op for n in g.nodes[-len(ops) :]] != list(ops):
            return False
        if g.output.ids != [len(g.nodes) - 1]:
            return False
        if g.output.kind == IDX_NONE:
            return True
        idx = np.asarray(g.output.array).reshape(-1)
        return np.array_equal(idx, np.arange(g.nodes[-1].num_folds))
