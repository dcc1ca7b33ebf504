"""Evaluation plans: the data the HIP backend consumes.

A *plan* is the folded, optimised layer list of a compiled circuit -- exactly the
information the reference keeps in ``TorchCircuit.address_book`` plus each layer's
``config``/``scope_idx``/parameter graph -- stripped of every Python object so it can be
serialised (JSON + int arrays) and shipped to a box where the reference is absent.

Reference structures this mirrors (paths relative to the reference checkout):

* ``AddressBookEntry(module, in_module_ids, in_fold_idx)``  cirkit/backend/torch/graph/modules.py:85-103
* ``LayerAddressBook.lookup`` (how the entry is consumed)     cirkit/backend/torch/circuits.py:30-71
* ``build_address_book_stacked_entry`` (index / unsqueeze)    cirkit/backend/torch/graph/folding.py:202-243
* ``ParameterAddressBook.lookup`` (parameter graphs)          cirkit/backend/torch/parameters/parameter.py:37-62

`plan_from_torch_circuit` is the drop-in boundary "b4" of SURVEY.md section 8(b): it walks a
compiled reference circuit by duck-typing (it never imports ``cirkit``), so it works on the
objects ``cirkit.pipeline.compile()`` returns.
"""

from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Any, Iterable, Mapping, Sequence

import numpy as np

PLAN_VERSION = 1

# fold-index kinds (folding.py:234-241)
IDX_ARRAY = "array"  # explicit (F, H) int64 gather over the fold-concatenation of the inputs
IDX_UNSQ0 = "unsq0"  # ``x[None]``      : all input folds become the arity axis of ONE fold
IDX_UNSQ1 = "unsq1"  # ``x[:, None]``   : identity over folds, arity 1
IDX_NONE = "none"  # ``x[()]``        : identity (parameter graphs only)

LAYER_TYPES = {
    "TorchCategoricalLayer": "categorical",
    "TorchGaussianLayer": "gaussian",
    "TorchBinomialLayer": "binomial",
    "TorchEmbeddingLayer": "embedding",
    "TorchConstantValueLayer": "constant",
    "TorchSumLayer": "sum",
    "TorchHadamardLayer": "hadamard",
    "TorchKroneckerLayer": "kronecker",
    "TorchCPTLayer": "cpt",
    "TorchTensorDotLayer": "tensordot",
    "TorchTuckerLayer": "tucker",
}

PARAM_OPS = {
    "TorchTensorParameter": "tensor",
    "TorchPointerParameter": "pointer",
    "TorchSoftmaxParameter": "softmax",
    "TorchLogSoftmaxParameter": "log_softmax",
    "TorchSigmoidParameter": "sigmoid",
    "TorchScaledSigmoidParameter": "scaled_sigmoid",
    "TorchExpParameter": "exp",
    "TorchLogParameter": "log",
    "TorchSquareParameter": "square",
    "TorchClampParameter": "clamp",
    "TorchSumParameter": "sum",
    "TorchHadamardParameter": "hadamard",
    "TorchKroneckerParameter": "kronecker",
    "TorchOuterProductParameter": "outer_product",
    "TorchReduceSumParameter": "reduce_sum",
    "TorchReduceProductParameter": "reduce_prod",
    "TorchReduceLSEParameter": "reduce_lse",
    "TorchOuterSumParameter": "outer_sum",
    "TorchIndexParameter": "index",
    "TorchSoftplusParameter": "softplus",
    "TorchConjugateParameter": "conj",
    "TorchMixingWeightParameter": "mixing_weight",
    "TorchMatMulParameter": "matmul",
    "TorchEinsumParameter": "einsum",
    "TorchFlattenParameter": "flatten",
    "TorchGaussianProductLogPartition": "gaussian_product_log_partition",
    "TorchGaussianProductMean": "gaussian_product_mean",
    "TorchGaussianProductStddev": "gaussian_product_stddev",
}


@dataclass
class FoldIndex:
    """How the inputs of a folded module are gathered from earlier modules' outputs."""

    ids: list[int]  # distinct producer module ids, concatenated along folds in this order
    kind: str  # IDX_*
    array: np.ndarray | None = None  # int64, for IDX_ARRAY

    def to_json(self, arrays: dict[str, np.ndarray], key: str) -> dict[str, Any]:
        d: dict[str, Any] = {"ids": list(self.ids), "kind": self.kind}
        if self.kind == IDX_ARRAY:
            assert self.array is not None
            arrays[key] = np.asarray(self.array, dtype=np.int64)
            d["array"] = key
        return d

    @staticmethod
    def from_json(d: Mapping[str, Any], arrays: Mapping[str, np.ndarray]) -> "FoldIndex":
        arr = np.asarray(arrays[d["array"]], dtype=np.int64) if d["kind"] == IDX_ARRAY else None
        return FoldIndex(list(d["ids"]), d["kind"], arr)


@dataclass
class ParamNode:
    op: str  # PARAM_OPS value
    num_folds: int
    shape: tuple[int, ...]  # per-fold output shape
    config: dict[str, Any] = field(default_factory=dict)
    inputs: list[FoldIndex] = field(default_factory=list)  # one per operand


@dataclass
class ParamGraph:
    """A layer parameter: small DAG re-evaluated by the reference on every forward
    (cirkit/backend/torch/parameters/parameter.py:180-188)."""

    nodes: list[ParamNode]
    output: FoldIndex
    num_folds: int
    shape: tuple[int, ...]

    @property
    def ops(self) -> list[str]:
        return [n.op for n in self.nodes]


@dataclass
class LayerSpec:
    type: str  # LAYER_TYPES value
    num_folds: int
    arity: int
    num_input_units: int
    num_output_units: int
    config: dict[str, Any]
    params: dict[str, ParamGraph]
    inputs: FoldIndex | None = None  # None for input layers
    scope_idx: np.ndarray | None = None  # (F, D') int64, input layers only


@dataclass
class Plan:
    semiring: str
    num_variables: int
    layers: list[LayerSpec]
    output: FoldIndex
    # name -> (shape incl. fold axis, dtype string); values live outside the plan
    tensors: dict[str, tuple[tuple[int, ...], str]] = field(default_factory=dict)
    name: str = ""

    # ------------------------------------------------------------------ sizes
    @property
    def num_params(self) -> int:
        return int(sum(int(np.prod(s)) for s, _ in self.tensors.values()))

    def algorithmic_bytes(self, batch: int, x_itemsize: int = 8) -> dict[str, float]:
        """every folded layer reads its logical inputs once and writes its
        output once, parameters once per forward, gather copies not counted."""
        esz = 8 if self.semiring == "complex-lse-sum" else 4
        rd = wr = 0
        for l in self.layers:
            if l.inputs is not None:
                rd += l.num_folds * l.arity * batch * l.num_input_units * esz
            elif l.scope_idx is not None and l.scope_idx.size:
                rd += int(l.scope_idx.size) * batch * x_itemsize
            b = 1 if (l.type == "constant" and self.num_variables == 0) else batch
            wr += l.num_folds * b * l.num_output_units * esz
        pb = sum(int(np.prod(s)) * (8 if "complex" in dt else 4) for s, dt in self.tensors.values())
        return {"read": float(rd), "write": float(wr), "params": float(pb), "total": float(rd + wr + pb)}

    # ------------------------------------------------------------------ (de)serialisation
    def to_json(self) -> tuple[dict[str, Any], dict[str, np.ndarray]]:
        arrays: dict[str, np.ndarray] = {}
        layers = []
        for i, l in enumerate(self.layers):
            d: dict[str, Any] = {
                "type": l.type,
                "F": l.num_folds,
                "H": l.arity,
                "Ki": l.num_input_units,
                "Ko": l.num_output_units,
                "config": l.config,
                "params": {},
            }
            if l.inputs is not None:
                d["inputs"] = l.inputs.to_json(arrays, f"L{i}_idx")
            if l.scope_idx is not None:
                arrays[f"L{i}_scope"] = np.asarray(l.scope_idx, dtype=np.int64)
                d["scope"] = f"L{i}_scope"
            for pn, pg in l.params.items():
                nodes = []
                for j, n in enumerate(pg.nodes):
                    nodes.append(
                        {
                            "op": n.op,
                            "F": n.num_folds,
                            "shape": list(n.shape),
                            "config": n.config,
                            "inputs": [
                                fi.to_json(arrays, f"L{i}_{pn}_n{j}_i{k}")
                                for k, fi in enumerate(n.inputs)
                            ],
                        }
                    )
                d["params"][pn] = {
                    "nodes": nodes,
                    "output": pg.output.to_json(arrays, f"L{i}_{pn}_out"),
                    "F": pg.num_folds,
                    "shape": list(pg.shape),
                }
            layers.append(d)
        doc = {
            "version": PLAN_VERSION,
            "name": self.name,
            "semiring": self.semiring,
            "num_variables": self.num_variables,
            "layers": layers,
            "output": self.output.to_json(arrays, "out_idx"),
            "tensors": {k: {"shape": list(s), "dtype": dt} for k, (s, dt) in self.tensors.items()},
        }
        return doc, arrays

    @staticmethod
    def from_json(doc: Mapping[str, Any], arrays: Mapping[str, np.ndarray]) -> "Plan":
        if doc.get("version") != PLAN_VERSION:
            raise ValueError(f"Unsupported plan version {doc.get('version')}")
        layers = []
        for d in doc["layers"]:
            params = {}
            for pn, pd in d["params"].items():
                nodes = [
                    ParamNode(
                        n["op"],
                        int(n["F"]),
                        tuple(n["shape"]),
                        dict(n["config"]),
                        [FoldIndex.from_json(fi, arrays) for fi in n["inputs"]],
                    )
                    for n in pd["nodes"]
                ]
                params[pn] = ParamGraph(
                    nodes, FoldIndex.from_json(pd["output"], arrays), int(pd["F"]), tuple(pd["shape"])
                )
            layers.append(
                LayerSpec(
                    d["type"],
                    int(d["F"]),
                    int(d["H"]),
                    int(d["Ki"]),
                    int(d["Ko"]),
                    dict(d["config"]),
                    params,
                    FoldIndex.from_json(d["inputs"], arrays) if "inputs" in d else None,
                    np.asarray(arrays[d["scope"]], dtype=np.int64) if "scope" in d else None,
                )
            )
        return Plan(
            doc["semiring"],
            int(doc["num_variables"]),
            layers,
            FoldIndex.from_json(doc["output"], arrays),
            {k: (tuple(v["shape"]), v["dtype"]) for k, v in doc["tensors"].items()},
            doc.get("name", ""),
        )

    def save(self, path_prefix: str) -> None:
        doc, arrays = self.to_json()
        with open(path_prefix + ".json", "w", encoding="utf-8") as f:
            json.dump(doc, f, separators=(",", ":"))
        np.savez_compressed(path_prefix + ".npz", **arrays)

    @staticmethod
    def load(path_prefix: str) -> "Plan":
        with open(path_prefix + ".json", encoding="utf-8") as f:
            doc = json.load(f)
        with np.load(path_prefix + ".npz") as z:
            arrays = {k: z[k] for k in z.files}
        return Plan.from_json(doc, arrays)


# ---------------------------------------------------------------------------------------------
# Resolving fold indices to (producer, fold) pairs -- shared by the HIP executor and the oracle
# ---------------------------------------------------------------------------------------------
def resolve_fold_index(fi: FoldIndex, folds_of: Sequence[int]) -> np.ndarray:
    """Expand a FoldIndex into an int64 array ``(..., 2)`` of ``(producer id, fold in producer)``.

    ``folds_of[i]`` is the fold count of module ``i``.  The leading shape is ``(F, H)`` for layers
    and ``(F,)`` for parameter-graph operands / outputs.  Mirrors the cumulative-offset rule of
    folding.py:213-222 read backwards.
    """
    sizes = [int(folds_of[i]) for i in fi.ids]
    total = int(sum(sizes))
    owner = np.concatenate([np.full(s, i, dtype=np.int64) for i, s in zip(fi.ids, sizes)])
    local = np.concatenate([np.arange(s, dtype=np.int64) for s in sizes])
    if fi.kind == IDX_ARRAY:
        assert fi.array is not None
        idx = np.asarray(fi.array, dtype=np.int64)
        if idx.size and (idx.min() < 0 or idx.max() >= total):
            raise ValueError("fold index out of range of the concatenated inputs")
    elif fi.kind == IDX_UNSQ0:
        idx = np.arange(total, dtype=np.int64)[None, :]
    elif fi.kind == IDX_UNSQ1:
        idx = np.arange(total, dtype=np.int64)[:, None]
    elif fi.kind == IDX_NONE:
        idx = np.arange(total, dtype=np.int64)
    else:
        raise ValueError(f"unknown fold index kind {fi.kind!r}")
    return np.stack([owner[idx], local[idx]], axis=-1)


# ---------------------------------------------------------------------------------------------
# Extraction from a compiled reference circuit (duck-typed)
# ---------------------------------------------------------------------------------------------
def _fold_index_of(ids: Sequence[int], idx: Any) -> FoldIndex:
    ids = [int(i) for i in ids]
    if isinstance(idx, tuple):
        if idx == ():
            return FoldIndex(ids, IDX_NONE)
        if idx == (None,):
            return FoldIndex(ids, IDX_UNSQ0)
        if len(idx) == 2 and idx[0] == slice(None) and idx[1] is None:
            return FoldIndex(ids, IDX_UNSQ1)
        raise ValueError(f"unrecognised fold index shortcut {idx!r}")
    arr = np.asarray(idx.detach().cpu().numpy() if hasattr(idx, "detach") else idx, dtype=np.int64)
    return FoldIndex(ids, IDX_ARRAY, arr)


def _jsonable(v: Any) -> Any:
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, (tuple, list)):
        return [_jsonable(i) for i in v]
    if isinstance(v, np.generic):
        return v.item()
    raise TypeError(f"cannot serialise config value {v!r}")


class _TensorTable:
    """Names the ``nn.Parameter`` owners so pointer nodes (cirkit .../nodes.py:223-279) in *other*
    circuits resolve to the same storage."""

    def __init__(self) -> None:
        self.by_id: dict[int, str] = {}
        self.values: dict[str, Any] = {}
        self.meta: dict[str, tuple[tuple[int, ...], str]] = {}

    def name_of(self, tensor_param: Any) -> str:
        key = id(tensor_param)
        if key not in self.by_id:
            name = f"t{len(self.by_id)}"
            self.by_id[key] = name
            t = tensor_param()  # (F, *shape) nn.Parameter
            self.values[name] = t.detach()
            self.meta[name] = (tuple(int(s) for s in t.shape), str(t.dtype).replace("torch.", ""))
        return self.by_id[key]


def _param_graph_of(p: Any, table: _TensorTable) -> ParamGraph:
    nodes: list[ParamNode] = []
    output: FoldIndex | None = None
    for entry in p.address_book:
        node = entry.module
        if node is None:
            output = _fold_index_of(entry.in_module_ids[0], entry.in_fold_idx[0])
            break
        cls = type(node).__name__
        if cls not in PARAM_OPS:
            raise NotImplementedError(
                f"parameter node {cls} is outside the supported hot path (SURVEY.md section 8 a13)"
            )
        op = PARAM_OPS[cls]
        cfg: dict[str, Any] = {}
        if op == "tensor":
            cfg["tensor"] = table.name_of(node)
        elif op == "pointer":
            cfg["tensor"] = table.name_of(node.deref())
            cfg["fold_idx"] = node.fold_idx
        else:
            for k, v in node.config.items():
                if k.startswith("in_shape") or k == "in_shapes":
                    continue
                cfg[k] = _jsonable(v)
        inputs = [
            _fold_index_of(mids, idx) for mids, idx in zip(entry.in_module_ids, entry.in_fold_idx)
        ]
        nodes.append(ParamNode(op, int(node.num_folds), tuple(int(s) for s in node.shape), cfg, inputs))
    if output is None:
        raise ValueError("malformed parameter address book (no output entry)")
    return ParamGraph(nodes, output, int(p.num_folds), tuple(int(s) for s in p.shape))


def plan_from_torch_circuit(
    circuit: Any, *, table: _TensorTable | None = None, name: str = ""
) -> tuple[Plan, dict[str, Any]]:
    """Walk a compiled (folded or not) reference ``TorchCircuit`` and return ``(plan, tensors)``.

    ``tensors`` maps the plan's tensor names to the circuit's ``nn.Parameter`` data (torch tensors
    that *share storage* with the reference circuit).
    """
    table = table or _TensorTable()
    layers: list[LayerSpec] = []
    output: FoldIndex | None = None
    semiring = None
    for entry in circuit.address_book:
        layer = entry.module
        if layer is None:
            output = _fold_index_of(entry.in_module_ids[0], entry.in_fold_idx[0])
            break
        # by class name along the MRO: a subclass of a reference layer (cirkit_amd/cirkit_plugin.py) is its base
        cls = next((k.__name__ for k in type(layer).__mro__ if k.__name__ in LAYER_TYPES), type(layer).__name__)
        if cls not in LAYER_TYPES:
            raise NotImplementedError(
                f"layer {cls} is outside the supported hot path (SURVEY.md section 8 a)"
            )
        semiring = layer.semiring
        params = {pn: _param_graph_of(p, table) for pn, p in layer.params.items()}
        cfg = {k: _jsonable(v) for k, v in layer.config.items()}
        spec = LayerSpec(
            LAYER_TYPES[cls],
            int(layer.num_folds),
            int(layer.arity),
            int(layer.num_input_units),
            int(layer.num_output_units),
            cfg,
            params,
        )
        if entry.in_module_ids:
            spec.inputs = _fold_index_of(entry.in_module_ids[0], entry.in_fold_idx[0])
        else:
            spec.scope_idx = np.asarray(layer.scope_idx.detach().cpu().numpy(), dtype=np.int64)
        layers.append(spec)
    if output is None:
        raise ValueError("malformed address book (no output entry)")
    sname = {
        "LSESumSemiring": "lse-sum",
        "ComplexLSESumSemiring": "complex-lse-sum",
        "SumProductSemiring": "sum-product",
    }[getattr(semiring, "__name__", type(semiring).__name__)]
    scope = getattr(circuit, "scope", None)
    nvars = (max(scope) + 1) if scope else 0
    used = set()
    for l in layers:
        for pg in l.params.values():
            for n in pg.nodes:
                if n.op in ("tensor", "pointer"):
                    used.add(n.config["tensor"])
    plan = Plan(
        sname,
        int(nvars),
        layers,
        output,
        {k: table.meta[k] for k in table.meta if k in used},
        name,
    )
    return plan, {k: table.values[k] for k in plan.tensors}


def tensor_table() -> _TensorTable:
    """A shared tensor table for extracting several circuits that point at each other's weights."""
    return _TensorTable()
