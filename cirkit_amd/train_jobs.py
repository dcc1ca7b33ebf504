"""The training step as LEVEL launches over JOBS, for circuits made of 64-unit dense / CP-T / mixing / Hadamard layers.

The reference trains by autograd through its layer-by-layer forward (notebooks/learning-a-circuit.ipynb cell 18 over
layers/inner.py:126-127, 266-273, optimized.py:171-178, semiring.py:383-408, parameters/nodes.py:764-772, 847-862).  This
module is the host side of cirkit_amd/csrc/ck_jobs.hip: it turns a folded plan into

* SUM jobs -- one fold of a dense / CP-T layer (64 -> 64 units): input = the sum of a list of blocks (a Hadamard product in
  log space: the product layers are never evaluated, their folds are LISTS), output one block, backward one gradient block;
* MIX jobs -- one fold of a mixing layer over H slots (each a list of blocks); a collapsed Sum -> Sum pair (a MatMul weight,
  nodes.py:802-805) is evaluated as what it was before the reference's optimizer collapsed it: a MIX job feeding a SUM job;
* NSUM jobs -- products that are kept (more than `MAX_LIST` factors) and gradients that several jobs read;
* the ROOT launch -- scalar sum folds + the final mixing layer + the log-likelihood sum + their backward;

orders them in levels (one launch per kind and level), and records the whole step -- parameter prologue, input layers, forward
levels, root, backward levels, input-layer backward -- as ONE native launch list per batch size (`ck_program`).  Every gradient
block has one writer; readers add the blocks of their list.  Parameter gradients leave the job epilogues as d theta (the softmax
behind every weight is differentiated by the workgroup that holds dW).  `HipTrainer` owns the buffers, the optimizer and the
collective; `JobStep.applies(trainer)` says why a plan does not take this form (then the layer-wise launch list runs)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi as capi
from .layers import HipCategoricalLayer, HipCPTLayer, HipGaussianLayer, HipHadamardLayer, HipSumLayer, HipTuckerLayer
from .plan import resolve_fold_index

MAX_LIST = 4  # a product of more blocks than this is materialised by an NSUM job
K = 64


def _expr(g, j: int, f: int):
    """The value of fold f of node j of a parameter graph as a nested tuple (op, node, fold, *operands)."""
    n = g.nodes[j]
    if n.op == "tensor":
        return ("tensor", j, f, n.config["tensor"])
    folds = [m.num_folds for m in g.nodes]
    kids = []
    for fi in n.inputs:
        pr = resolve_fold_index(fi, [folds[i] if i in fi.ids else 0 for i in range(max(fi.ids) + 1)]).reshape(-1, 2)
        kids.append(_expr(g, int(pr[f, 0]), int(pr[f, 1])))
    return (n.op, j, f, *kids)


def _out_expr(g, f: int):
    folds = [m.num_folds for m in g.nodes]
    pr = resolve_fold_index(g.output, [folds[i] if i in g.output.ids else 0 for i in range(max(g.output.ids) + 1)]).reshape(-1, 2)
    return _expr(g, int(pr[f, 0]), int(pr[f, 1]))


def _is_softmax_of_tensor(g, e) -> bool:
    return (e[0] == "softmax" and e[3][0] == "tensor" and int(g.nodes[e[1]].config["dim"]) == len(g.nodes[e[1]].shape) - 1)


class JobStep:
    """Structure (independent of the batch size) + per-batch-size bindings of the job form of a training step."""

    def __init__(self, trainer) -> None:
        self.tr = trainer
        self.c = trainer.circuit
        self._bound: dict[int, dict] = {}
        self.why = self._analyse()

    # ---- analysis -------------------------------------------------------------------------------------------------------
    def _analyse(self) -> str | None:
        tr, c = self.tr, self.c
        plan = tr.plan
        if c._complex or len(c._out_pairs) != 1:
            return "needs a real circuit with one output"
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        n_layers = len(c.layers)
        vals: dict[tuple[int, int], list] = {}
        self.sum_jobs: list[dict] = []
        self.mix_jobs: list[dict] = []
        self.nsum_jobs: list[dict] = []
        self.inputs: list[int] = []
        self.n_extra = 0
        gsrc: dict[tuple, list] = {}
        producer: dict[tuple, dict] = {}  # block -> the job that writes it (forward)
        used: dict[tuple[str, int], str] = {}  # (tensor, fold) -> who differentiates it
        scalars: dict[tuple[int, int], dict] = {}

        def extra(n: int = 1) -> int:
            first = self.n_extra
            self.n_extra += n
            return first

        def claim(name: str, fold: int, who: str) -> bool:
            if (name, fold) in used:
                return False
            used[(name, fold)] = who
            return True

        def feed(blocks, gid) -> None:
            for x in blocks:
                gsrc.setdefault(x, []).append(gid)

        def level_of(blocks) -> int:
            return 1 + max((producer[x]["lf"] if x in producer else 0) for x in blocks)

        def gather(ch_f) -> list:
            lst: list = []
            for p, q in ch_f:
                lst += vals[(int(p), int(q))]
            return lst

        def shorten(lst: list) -> list:
            """A list of more than MAX_LIST blocks: materialise its sum (one NSUM job); the gradient of every member is the
            gradient of the sum."""
            if len(lst) <= MAX_LIST:
                return lst
            out = ("x", extra())
            job = {"ins": list(lst), "out": out, "lf": level_of(lst)}
            self.nsum_jobs.append(job)
            producer[out] = job
            feed(lst, ("ref", out))
            return [out]

        final_mix = None
        for i, (spec, l) in enumerate(zip(plan.layers, c.layers)):
            ch = c._children[i]
            F = l.num_folds
            if isinstance(l, HipCategoricalLayer) and type(l) is HipCategoricalLayer:
                if l.num_output_units != K or l.probs is None or l.probs.softmax_source() is None:
                    return f"layer {i}: Categorical layers need 64 units and probs = softmax(tensor)"
                name = l.probs.graph.nodes[0].config["tensor"]
                for f in range(F):
                    vals[(i, f)] = [("a", i, f)]
                    if not claim(name, f, f"layer {i}"):
                        return f"tensor {name} is shared"
                self.inputs.append(i)
            elif isinstance(l, HipGaussianLayer):
                if l.num_output_units != K or l.log_partition is not None or (set(l.mean.ops) | set(l.stddev.ops)) - tr._PARAM_OPS:
                    return f"layer {i}: Gaussian layers need 64 units, no log-partition and plain parameters"
                for f in range(F):
                    vals[(i, f)] = [("a", i, f)]
                self.inputs.append(i)
            elif isinstance(l, HipHadamardLayer):
                for f in range(F):
                    lst = gather(ch[f])
                    if len(lst) <= MAX_LIST:
                        vals[(i, f)] = lst  # virtual: the product is the list
                    else:
                        out = ("a", i, f)
                        job = {"ins": lst, "out": out, "lf": level_of(lst)}
                        self.nsum_jobs.append(job)
                        producer[out] = job
                        feed(lst, ("ref", out))
                        vals[(i, f)] = [out]
            elif isinstance(l, (HipSumLayer, HipCPTLayer)) and not isinstance(l, HipTuckerLayer):
                Ki, Ko = l.num_input_units, l.num_output_units
                prod = l._mode == capi.CK_SUM_PROD or l.arity == 1
                if Ko == 1 and Ki == K and prod and l.weight.softmax_source() is not None:
                    name = l.weight.graph.nodes[0].config["tensor"]
                    for f in range(F):
                        if not claim(name, f, f"layer {i}"):
                            return f"tensor {name} is shared"
                        scalars[(i, f)] = {"layer": i, "fold": f, "ins": gather(ch[f]), "theta": (name, f)}
                elif Ko == 1 and Ki == 1 and l._mixing and l.weight.mixing_softmax_source() is not None and F == 1 and i == po:
                    name = l.weight.graph.nodes[0].config["tensor"]
                    if not claim(name, 0, f"layer {i}"):
                        return f"tensor {name} is shared"
                    kids = [(int(p), int(q)) for p, q in ch[0]]
                    if any(k not in scalars for k in kids) or len(set(kids)) != len(kids):
                        return "the final mixing layer must read distinct scalar sum folds"
                    final_mix = {"layer": i, "kids": kids, "theta": (name, 0)}
                elif Ki == K and Ko == K and l._mixing and l.weight.mixing_softmax_source() is not None:
                    if l.arity > 16:
                        return f"layer {i}: a mixing layer over more than 16 slots"
                    name = l.weight.graph.nodes[0].config["tensor"]
                    for f in range(F):
                        if not claim(name, f, f"layer {i}"):
                            return f"tensor {name} is shared"
                        self._add_mix(i, f, [vals[(int(p), int(q))] for p, q in ch[f]], ("a", i, f), ("layer", i, f), (name, f),
                                      shorten, level_of, producer, feed, extra)
                        vals[(i, f)] = [("a", i, f)]
                elif Ki == K and Ko == K and prod and l.weight.softmax_source() is not None:
                    name = l.weight.graph.nodes[0].config["tensor"]
                    for f in range(F):
                        if not claim(name, f, f"layer {i}"):
                            return f"tensor {name} is shared"
                        ins = shorten(gather(ch[f]))
                        job = {"layer": i, "fold": f, "ins": ins, "out": ("a", i, f), "gx": ("x", extra()), "w": ("layer", i, f),
                               "theta": (name, f), "lf": level_of(ins)}
                        self.sum_jobs.append(job)
                        producer[job["out"]] = job
                        feed(ins, job["gx"])
                        vals[(i, f)] = [job["out"]]
                elif Ki == K and Ko == K and l._mode == capi.CK_SUM_CAT and l.arity > 1 and not l._mixing:
                    # per fold: a mixing weight, or a dense weight times a mixing weight (the collapsed pair)
                    g = l.weight.graph
                    if l.arity > 16:
                        return f"layer {i}: more than 16 slots"
                    for f in range(F):
                        e = _out_expr(g, f)
                        slots = [vals[(int(p), int(q))] for p, q in ch[f]]
                        if e[0] == "mixing_weight" and _is_softmax_of_tensor(g, e[3]):
                            sm = e[3]
                            if not claim(sm[3][3], sm[3][2], f"layer {i}"):
                                return f"tensor {sm[3][3]} is shared"
                            self._add_mix(i, f, slots, ("a", i, f), ("node", i, sm[1], sm[2]), (sm[3][3], sm[3][2]),
                                          shorten, level_of, producer, feed, extra)
                        elif (e[0] == "matmul" and _is_softmax_of_tensor(g, e[3]) and e[4][0] == "mixing_weight"
                              and _is_softmax_of_tensor(g, e[4][3])):
                            sd, sm = e[3], e[4][3]
                            if not claim(sd[3][3], sd[3][2], f"layer {i}") or not claim(sm[3][3], sm[3][2], f"layer {i}"):
                                return f"tensors of layer {i} are shared"
                            mid = ("x", extra())
                            self._add_mix(i, f, slots, mid, ("node", i, sm[1], sm[2]), (sm[3][3], sm[3][2]),
                                          shorten, level_of, producer, feed, extra)
                            job = {"layer": i, "fold": f, "ins": [mid], "out": ("a", i, f), "gx": ("x", extra()),
                                   "w": ("node", i, sd[1], sd[2]), "theta": (sd[3][3], sd[3][2]), "lf": level_of([mid])}
                            self.sum_jobs.append(job)
                            producer[job["out"]] = job
                            feed([mid], job["gx"])
                        else:
                            return f"layer {i}: weight parameterisation {l.weight.ops}"
                        vals[(i, f)] = [("a", i, f)]
                else:
                    return f"layer {i}: a {spec.type} layer of {Ki} -> {Ko} units, arity {l.arity}, weight {l.weight.ops}"
            else:
                return f"layer {i}: layer type {spec.type!r}"
        # (tensor folds nobody reads keep a zero gradient: the flat gradient buffer starts as zeros and only claimed folds are written)
        # the root: the scalar folds in the order the final mixing layer reads them
        if final_mix is not None:
            order = final_mix["kids"]
            if set(order) != set(scalars):
                return "scalar sum folds outside the final mixing layer"
        else:
            if len(scalars) != 1 or (po, fo) not in scalars:
                return "the circuit must end in a scalar sum fold or a final mixing layer over scalar sum folds"
            order = [(po, fo)]
        if len(order) > 16:
            return "more than 16 scalar folds under the final mixing layer"
        g0 = extra(len(order))
        self.root = {"folds": [scalars[k] for k in order], "mix": final_mix, "gx0": g0}
        for r, k in enumerate(order):
            feed(scalars[k]["ins"], ("x", g0 + r))
        if not self.sum_jobs:
            return "no 64-unit sum layer"

        # gradient lists: expand references to kept products, then materialise lists that several readers share
        memo: dict[tuple, tuple] = {}

        def sources(x) -> tuple:
            if x in memo:
                return memo[x]
            out: list = []
            for gsid in gsrc.get(x, []):
                if gsid[0] == "ref":
                    out += list(sources(gsid[1]))
                else:
                    out.append(gsid)
            memo[x] = tuple(out)
            return memo[x]

        readers: dict[tuple, int] = {}
        wanted = [j["out"] for j in self.sum_jobs + self.mix_jobs] + [("a", i, f) for i in self.inputs for f in range(c.layers[i].num_folds)]
        for x in wanted:
            readers[sources(x)] = readers.get(sources(x), 0) + 1
        self.gsum_jobs: list[dict] = []
        shared: dict[tuple, tuple] = {}
        for lst, n in readers.items():
            if len(lst) >= 3 and n >= 2:
                out = ("x", extra())
                shared[lst] = out
                self.gsum_jobs.append({"ins": list(lst), "out": out})
        self._sources = lambda x: ((shared[sources(x)],) if sources(x) in shared else sources(x))
        for j in self.sum_jobs + self.mix_jobs:
            j["g"] = list(self._sources(j["out"]))
            if not j["g"]:
                return f"layer {j['layer']} fold {j['fold']} feeds nothing"
        # backward levels: the root is level 0; a job follows the writers of its gradient list
        writer: dict[tuple, dict] = {}
        for r in range(len(order)):
            writer[("x", g0 + r)] = {"lb": 0}
        for j in self.sum_jobs:
            writer[j["gx"]] = j
        for j in self.mix_jobs:
            for h in range(j["H"]):
                writer[("x", j["gx0"] + h)] = j
        for j in self.gsum_jobs:
            writer[j["out"]] = j

        def lb(j: dict) -> int:
            if "lb" not in j:
                j["lb"] = 1 + max(lb(writer[gsid]) for gsid in (j["g"] if "g" in j else j["ins"]))
            return j["lb"]

        for j in self.sum_jobs + self.mix_jobs + self.gsum_jobs:
            lb(j)
        # the gradient of every input-layer fold, gathered into a contiguous (F, B, 64) block per layer for its backward
        self.input_g: dict[int, dict] = {}
        for i in self.inputs:
            Fi = c.layers[i].num_folds
            first = extra(Fi)
            lists = [list(self._sources(("a", i, f))) for f in range(Fi)]
            if any(not lst for lst in lists):
                return f"input layer {i} has a fold nobody reads"
            self.input_g[i] = {"first": first, "lists": lists,
                               "lb": 1 + max(lb(writer[gsid]) for lst in lists for gsid in lst)}
        return None

    def _add_mix(self, i, f, slots, out, w, theta, shorten, level_of, producer, feed, extra) -> None:
        S = max(len(s) for s in slots)
        if any(len(s) != S for s in slots):  # uniform slot length: longer products are kept
            slots = [shorten(s) if len(s) > 1 else s for s in slots]
            S = max(len(s) for s in slots)
            if any(len(s) != S for s in slots):
                raise NotImplementedError("mixing slots that are products of different numbers of blocks")
        H = len(slots)
        flat = [x for s in slots for x in s]
        job = {"layer": i, "fold": f, "slots": slots, "H": H, "S": S, "out": out, "gx0": extra(H), "w": w, "theta": theta,
               "lf": level_of(flat)}
        self.mix_jobs.append(job)
        producer[out] = job
        for h, s in enumerate(slots):
            feed(s, ("x", job["gx0"] + h))

    # ---- binding ----------------------------------------------------------------------------------------------------------
    def _weight_ptr(self, w) -> int:
        c = self.c
        if w[0] == "layer":
            l = c.layers[w[1]]
            return l._w.data_ptr() + w[2] * (l._w.numel() // l._w.shape[0]) * 4
        _, i, node, fold = w  # the evaluated softmax node of a parameter graph that `prepare` evaluates node by node
        t = c.layers[i].weight._last_outs[node]
        return t.data_ptr() + fold * (t.numel() // t.shape[0]) * 4

    def bind(self, B: int) -> dict:
        tr, c = self.tr, self.c
        bd = c._bind(B)
        st = self._bound.get(B)
        if st is not None and st["arena_ptr"] == bd.arena.data_ptr() and st["store_version"] == c.store.version:
            return st
        if st is not None:
            capi.load().ck_program_destroy(st["prog"])
        dev = c.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            c._enqueue_params(stream)  # (allocates every derived-parameter buffer the tables below point at)
            torch.cuda.synchronize(dev)
        n_cu = c._n_cu
        blk = B * K
        extra = torch.zeros(max(1, self.n_extra) * blk, dtype=torch.float32, device=dev)
        x0 = extra.data_ptr()

        def addr(x) -> int:
            if x[0] == "a":
                v = bd.views[x[1]]
                return v.data_ptr() + x[2] * B * v.shape[2] * 4
            return x0 + x[1] * blk * 4

        pool: list[int] = []

        def put(blocks) -> tuple[int, int]:
            off = len(pool)
            pool.extend(addr(x) for x in blocks)
            return off, len(blocks)

        grads, flat_g, flat_p = tr.grads, tr._flat_grad, tr._flat_param

        def grad_ptr(theta) -> int:
            t = grads[theta[0]]
            return t.data_ptr() + theta[1] * (t.numel() // t.shape[0]) * 4

        def theta_ptrs(theta) -> tuple[int, int, int]:
            t = c.store[theta[0]]
            p = t.data_ptr() + theta[1] * (t.numel() // t.shape[0]) * 4
            off = p - flat_p.data_ptr()
            m1 = tr._m1.data_ptr() + off if tr._m1 is not None else 0
            m2 = tr._m2.data_ptr() + off if tr._m2 is not None else 0
            return p, m1, m2

        tiles = (B + 31) // 32
        parts: list[torch.Tensor] = []
        keep: list[torch.Tensor] = [extra]

        def splits_for(n_jobs: int, max_split: int) -> int:
            return int(max(1, min(max_split, -(-2 * n_cu // max(1, n_jobs)))))

        def sum_table(jobs: list[dict], backward: bool) -> tuple[torch.Tensor, int]:
            ns = splits_for(len(jobs), max(1, tiles // 4))
            rows_per = -(-tiles // ns) * 32
            ns = -(-B // rows_per)
            tab = np.zeros(len(jobs) * ns, dtype=np.dtype(capi.SUM_JOB_DTYPE))
            part = tick = None
            if backward and ns > 1:
                part = torch.zeros(len(jobs) * ns * 4096, dtype=torch.float32, device=dev)
                tick = torch.zeros(len(jobs), dtype=torch.int32, device=dev)
                keep.extend([part, tick])
            for n, j in enumerate(jobs):
                ioff, inum = put(j["ins"])
                goff, gnum = put(j["g"]) if backward else (0, 0)
                th, m1, m2 = theta_ptrs(j["theta"])
                w = self._weight_ptr(j["w"])
                for sp in range(ns):
                    r = tab[n * ns + sp]
                    r["w"], r["out"], r["gx"], r["dtheta"] = w, addr(j["out"]), addr(j["gx"]), grad_ptr(j["theta"])
                    r["theta"], r["m1"], r["m2"], r["w_out"] = th, m1, m2, w
                    r["in_off"], r["n_in"], r["g_off"], r["n_g"] = ioff, inum, goff, gnum
                    r["row0"], r["row1"] = sp * rows_per, min(B, (sp + 1) * rows_per)
                    r["split"], r["n_split"], r["mode"] = sp, ns, 1
                    if part is not None:
                        r["part"], r["ticket"] = part.data_ptr() + n * ns * 4096 * 4, tick.data_ptr() + n * 4
            t = torch.from_numpy(tab.view(np.uint8).reshape(len(tab), -1)).to(dev)
            keep.append(t)
            return t, len(tab)

        def mix_table(jobs: list[dict], backward: bool) -> tuple[torch.Tensor, int, int]:
            hmax = max(j["H"] for j in jobs)
            hpad = 2 if hmax <= 2 else 4 if hmax <= 4 else 8 if hmax <= 8 else 16
            ns = splits_for(len(jobs), max(1, B // 64))
            rows_per = -(-(-(-B // ns)) // 16) * 16
            ns = -(-B // rows_per)
            tab = np.zeros(len(jobs) * ns, dtype=np.dtype(capi.MIX_JOB_DTYPE))
            part = tick = None
            if backward and ns > 1:
                part = torch.zeros(len(jobs) * ns * K * hpad, dtype=torch.float32, device=dev)
                tick = torch.zeros(len(jobs), dtype=torch.int32, device=dev)
                keep.extend([part, tick])
            for n, j in enumerate(jobs):
                ioff, _ = put([x for s in j["slots"] for x in s])
                goff, gnum = put(j["g"]) if backward else (0, 0)
                th, m1, m2 = theta_ptrs(j["theta"])
                w = self._weight_ptr(j["w"])
                for sp in range(ns):
                    r = tab[n * ns + sp]
                    r["w"], r["out"], r["gx"], r["dtheta"] = w, addr(j["out"]), addr(("x", j["gx0"])), grad_ptr(j["theta"])
                    r["theta"], r["m1"], r["m2"], r["w_out"] = th, m1, m2, w
                    r["in_off"], r["H"], r["S"], r["g_off"], r["n_g"] = ioff, j["H"], j["S"], goff, gnum
                    r["row0"], r["row1"] = sp * rows_per, min(B, (sp + 1) * rows_per)
                    r["split"], r["n_split"], r["mode"] = sp, ns, 1
                    if part is not None:
                        r["part"], r["ticket"] = part.data_ptr() + n * ns * K * hpad * 4, tick.data_ptr() + n * 4
            t = torch.from_numpy(tab.view(np.uint8).reshape(len(tab), -1)).to(dev)
            keep.append(t)
            return t, len(tab), hmax

        def nsum_table(items: list[tuple[list, int]]) -> tuple[torch.Tensor, int]:
            tab = np.zeros(len(items), dtype=np.dtype(capi.NSUM_JOB_DTYPE))
            for r, (ins, out) in zip(tab, items):
                r["in_off"], r["n_in"] = put(ins)
                r["out"] = out
            t = torch.from_numpy(tab.view(np.uint8).reshape(len(tab), -1)).to(dev)
            keep.append(t)
            return t, len(tab)

        def by_level(jobs: list[dict], key: str) -> dict[int, list[dict]]:
            out: dict[int, list[dict]] = {}
            for j in jobs:
                out.setdefault(j[key], []).append(j)
            return out

        launches: list[tuple] = []  # (what, table, n, ...), in issue order
        fs, fm, fn = by_level(self.sum_jobs, "lf"), by_level(self.mix_jobs, "lf"), by_level(self.nsum_jobs, "lf")
        for lv in sorted(set(fs) | set(fm) | set(fn)):
            if lv in fn:
                launches.append(("nsum",) + nsum_table([(j["ins"], addr(j["out"])) for j in fn[lv]]))
            if lv in fs:
                launches.append(("sum_fwd",) + sum_table(fs[lv], False))
            if lv in fm:
                launches.append(("mix_fwd",) + mix_table(fm[lv], False))
        launches.append(("root",))
        bs, bm, bg = by_level(self.sum_jobs, "lb"), by_level(self.mix_jobs, "lb"), by_level(self.gsum_jobs, "lb")
        bi = {}
        for i, ig in self.input_g.items():
            bi.setdefault(ig["lb"], []).append(i)
        for lv in sorted(set(bs) | set(bm) | set(bg) | set(bi)):
            if lv in bg:
                launches.append(("nsum",) + nsum_table([(j["ins"], addr(j["out"])) for j in bg[lv]]))
            if lv in bs:
                launches.append(("sum_bwd",) + sum_table(bs[lv], True))
            if lv in bm:
                launches.append(("mix_bwd",) + mix_table(bm[lv], True))
            for i in bi.get(lv, []):
                ig = self.input_g[i]
                launches.append(("input_bwd", i) + nsum_table([(lst, x0 + (ig["first"] + f) * blk * 4) for f, lst in enumerate(ig["lists"])]))
        # the root launch
        root = self.root
        R = len(root["folds"])
        rin = np.zeros((2, R), dtype=np.int32)
        ptrs = np.zeros((6, R), dtype=np.uint64)  # w, dtheta, theta, m1, m2, w_out
        for r, sc in enumerate(root["folds"]):
            rin[0, r], rin[1, r] = put(sc["ins"])
            l = c.layers[sc["layer"]]
            w = l._w.data_ptr() + sc["fold"] * K * 4
            th, m1, m2 = theta_ptrs(sc["theta"])
            ptrs[:, r] = (w, grad_ptr(sc["theta"]), th, m1, m2, w)
        rin_d = torch.from_numpy(rin).to(dev)
        ptrs_d = torch.from_numpy(ptrs.view(np.int64)).to(dev)
        n_wg = int(max(1, min(64, (B + 3) // 4)))
        rpart = torch.zeros(n_wg * 1042, dtype=torch.float32, device=dev)
        rtick = torch.zeros(1, dtype=torch.int32, device=dev)
        seed = torch.zeros(B, dtype=torch.float32, device=dev)
        pool_d = torch.from_numpy(np.asarray(pool, dtype=np.uint64).view(np.int64)).to(dev)
        keep.extend([rin_d, ptrs_d, rpart, rtick, seed, pool_d])
        ra = capi.RootLaunch()
        ra.pool, ra.in_off, ra.n_in = pool_d.data_ptr(), rin_d[0].data_ptr(), rin_d[1].data_ptr()
        ra.w, ra.dtheta_w, ra.theta_w = ptrs_d[0].data_ptr(), ptrs_d[1].data_ptr(), ptrs_d[2].data_ptr()
        ra.m1_w, ra.m2_w, ra.w_out = ptrs_d[3].data_ptr(), ptrs_d[4].data_ptr(), ptrs_d[5].data_ptr()
        po, fo = int(c._out_pairs[0, 0]), int(c._out_pairs[0, 1])
        ra.out = bd.views[po][fo].data_ptr()
        ra.gx, ra.seed, ra.ll = x0 + root["gx0"] * blk * 4, seed.data_ptr(), bd.ll.data_ptr()
        ra.part, ra.ticket = rpart.data_ptr(), rtick.data_ptr()
        if root["mix"] is not None:
            lm = c.layers[root["mix"]["layer"]]
            th, m1, m2 = theta_ptrs(root["mix"]["theta"])
            ra.c, ra.dtheta_c = lm._w.data_ptr(), grad_ptr(root["mix"]["theta"])
            ra.theta_c, ra.m1_c, ra.m2_c, ra.c_out = th, m1, m2, lm._w.data_ptr()
        ra.opt, ra.bad_flag = None, (c._bad_input.data_ptr() if (c.validate_inputs and c._int_input) else None)
        ra.seed_const, ra.R, ra.B, ra.mode, ra.n_wg = 0.0, R, B, 1, n_wg
        st = {"arena_ptr": bd.arena.data_ptr(), "store_version": c.store.version, "keep": keep, "launches": launches, "root": ra,
              "pool": pool_d, "seed": seed, "seed_value": None, "extra": extra, "x0": x0, "prog": None, "dT": {}}
        st["prog"] = self._record(bd, st, B)
        while len(self._bound) >= 4:
            old = self._bound.pop(next(iter(self._bound)))
            capi.load().ck_program_destroy(old["prog"])
        self._bound[B] = st
        return st

    # ---- the launch list ----------------------------------------------------------------------------------------------------
    def _record(self, bd, st: dict, B: int):
        tr, c = self.tr, self.c
        prog = C.c_void_p()
        capi.call("ck_program_begin", C.byref(prog))
        try:
            self._enqueue(bd, st, B, 0)
        finally:
            capi.call("ck_program_end", prog)
        return prog

    def _enqueue(self, bd, st: dict, B: int, stream: int) -> None:
        tr, c = self.tr, self.c
        pool = st["pool"].data_ptr()
        blk = B * K
        c._enqueue_params(stream)  # every parameter graph, once per step (parameters/parameter.py:180-188)
        D = c.plan.num_variables
        for i in self.inputs:
            l = c.layers[i]
            l.launch_input(bd.xt if l.wants_float_input else bd.xt_i, D, bd.views[i], B, stream)
        for la in st["launches"]:
            what = la[0]
            if what == "nsum":
                capi.call("ck_jobs_nsum", la[1].data_ptr(), la[2], pool, blk, stream)
            elif what == "sum_fwd":
                capi.call("ck_jobs_sum64_fwd", la[1].data_ptr(), la[2], pool, stream)
            elif what == "mix_fwd":
                capi.call("ck_jobs_mix_fwd", la[1].data_ptr(), la[2], pool, la[3], stream)
            elif what == "root":
                capi.call("ck_jobs_root", C.byref(st["root"]), stream)
            elif what == "sum_bwd":
                capi.call("ck_jobs_sum64_bwd", la[1].data_ptr(), la[2], pool, None, stream)
            elif what == "mix_bwd":
                capi.call("ck_jobs_mix_bwd", la[1].data_ptr(), la[2], pool, la[3], blk, None, stream)
            elif what == "input_bwd":
                i = la[1]
                capi.call("ck_jobs_nsum", la[2].data_ptr(), la[3], pool, blk, stream)
                self._input_backward(i, bd, st, B, stream)

    def _input_backward(self, i: int, bd, st: dict, B: int, stream: int) -> None:
        """The backward of input layer i over its gathered (F, B, 64) gradient -- the launches of the layer-wise trainer."""
        tr, c = self.tr, self.c
        l = c.layers[i]
        g = st["x0"] + self.input_g[i]["first"] * B * K * 4
        dev = c.device
        if isinstance(l, HipCategoricalLayer):
            dT = st["dT"].get(i)
            if dT is None:
                dT = st["dT"][i] = torch.zeros((l.num_folds, l.num_categories + 1, K), dtype=torch.float32, device=dev)
            capi.call("ck_categorical_bwd", g, None, bd.xt_i.data_ptr(), l._scope(dev).data_ptr(), dT.data_ptr(), l.num_folds, B, K,
                      l.num_categories, 0, None, stream)
            name = l.probs.graph.nodes[0].config["tensor"]
            capi.call("ck_param_log_table_bwd", l._table.data_ptr(), dT.data_ptr(), tr.grads[name].data_ptr(), l.num_folds, K,
                      l.num_categories, 0, stream)
        else:  # Gaussian
            mean, stddev, _ = l._vals
            dm = st["dT"].get((i, "m"))
            if dm is None:
                dm = st["dT"][(i, "m")] = torch.zeros_like(mean)
                st["dT"][(i, "s")] = torch.zeros_like(stddev)
            ds = st["dT"][(i, "s")]
            capi.call("ck_gaussian_bwd", g, bd.xt.data_ptr(), l._scope(dev).data_ptr(), mean.data_ptr(), stddev.data_ptr(), dm.data_ptr(),
                      ds.data_ptr(), l.num_folds, B, K, stream)
            for p in (l.mean, l.stddev):  # (the parameter backward ADDS into the tensors' gradients)
                for n in p.graph.nodes:
                    if n.op == "tensor":
                        t = tr.grads[n.config["tensor"]]
                        capi.call("ck_fill_f32", t.data_ptr(), t.numel(), 0.0, stream)
            l.mean.backward(dm, tr.grads, stream)
            l.stddev.backward(ds, tr.grads, stream)

    # ---- one step -------------------------------------------------------------------------------------------------------------
    def loss_and_grads(self, x: torch.Tensor, gB: float) -> torch.Tensor:
        """Forward + backward of ``-(1 / gB) sum_b log p(x_b)`` over the recorded launch list; returns the circuit's
        [sum log p, rows] pair (device, overwritten by the next call at this batch size)."""
        c = self.c
        B = int(x.shape[0])
        st = self.bind(B)
        bd = c._bind(B)
        with torch.cuda.device(c.device):
            stream = torch.cuda.current_stream(c.device).cuda_stream
            if st["seed_value"] != -1.0 / gB:
                capi.call("ck_fill_f32", st["seed"].data_ptr(), B, -1.0 / gB, stream)
                st["seed_value"] = -1.0 / gB
            xf, xi = c._prepare_input(x)
            c._stage_input(bd, xf, xi, stream)
            capi.call("ck_program_launch", st["prog"], 0, stream)
        return bd.ll

    def num_launches(self, B: int) -> int:
        return int(capi.load().ck_program_num_ops(self.bind(B)["prog"]))
