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
import os

import numpy as np
import torch

from . import _capi as capi
from .layers import HipCategoricalLayer, HipCPTLayer, HipGaussianLayer, HipHadamardLayer, HipSumLayer, HipTuckerLayer
from .plan import resolve_fold_index

MAX_LIST = 4  # a product of more blocks than this is materialised by an NSUM job
K = 64
FOLD_MIX_BWD = os.environ.get("CK_JOBS_FOLD_MIX", "1") != "0"  # (lab switch: 0 keeps a backward launch per mixing level)


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
        self._opt: torch.Tensor | None = None
        self._opt_key = None
        self._own_state = None  # the store's state after this object's last in-place update (fused optimizer)
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
        self.gauss: dict[int, list[dict]] = {}  # Gaussian layers whose folds are backward jobs
        self.cat: set[int] = set()  # Categorical layers whose folds are backward jobs
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
                if l.scope_idx.shape[1] == 1 and ((l.num_categories + 1) * 65 + 1280) * 4 + (2 * 16 * ((l.num_categories + 16) // 16) + 4100) * 4 <= 160 * 1024:
                    self.cat.add(i)  # its folds' backward (+ optimizer + next table) is one launch of jobs (`ck_jobs_cat_bwd`)
            elif isinstance(l, HipGaussianLayer):
                if l.num_output_units != K or l.log_partition is not None or (set(l.mean.ops) | set(l.stddev.ops)) - tr._PARAM_OPS:
                    return f"layer {i}: Gaussian layers need 64 units, no log-partition and plain parameters"
                for f in range(F):
                    vals[(i, f)] = [("a", i, f)]
                self.inputs.append(i)
                # mean = a tensor, stddev = a tensor or its scaled sigmoid (what the templates build): the fold's backward is a
                # job of its own (`ck_jobs_gauss_bwd`) reading its gradient list; anything else takes the layer-wise launches
                recs = []
                for f in range(F):
                    em, es = _out_expr(l.mean.graph, f), _out_expr(l.stddev.graph, f)
                    ss = es[0] == "scaled_sigmoid"
                    et = es[3] if ss else es
                    if em[0] != "tensor" or et[0] != "tensor" or l.scope_idx.shape[1] != 1:
                        recs = None
                        break
                    cfg = l.stddev.graph.nodes[es[1]].config if ss else {}
                    recs.append({"mean": (em[3], em[2]), "sd": (et[3], et[2]), "ss": ss, "vmin": float(cfg.get("vmin", 0.0)),
                                 "vmax": float(cfg.get("vmax", 1.0))})
                if recs is not None and all(claim(r["mean"][0], r["mean"][1], f"layer {i}") and claim(r["sd"][0], r["sd"][1], f"layer {i}")
                                            for r in recs):
                    self.gauss[i] = recs
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
                        scalars[(i, f)] = {"layer": i, "fold": f, "ins": shorten(gather(ch[f])), "theta": (name, f)}
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
        self.root = {"folds": [scalars[k] for k in order], "mix": final_mix, "gx0": g0, "zero": extra()}  # (a block nobody writes)
        for r, k in enumerate(order):
            feed(scalars[k]["ins"], ("x", g0 + r))
        if not self.sum_jobs:
            return "no 64-unit sum layer"

        # a Categorical layer whose folds are only read by sum jobs, each as the job's single input, is never evaluated: those jobs
        # gather the rows of its log-probability table themselves
        self.gathered: set[int] = set()
        if self.cat:
            seen: dict[int, bool] = {i: True for i in self.cat}
            def note(blocks, ok: bool) -> None:
                for x in blocks:
                    if x[0] == "a" and x[1] in seen and not ok:
                        seen[x[1]] = False
            for j in self.sum_jobs:
                note(j["ins"], len(j["ins"]) == 1)
            for j in self.mix_jobs:
                note([x for sl in j["slots"] for x in sl], False)
            for j in self.nsum_jobs:
                note(j["ins"], False)
            for k in order:
                note(scalars[k]["ins"], False)
            self.gathered = {i for i, ok in seen.items() if ok and (po != i)}
            for j in self.sum_jobs:
                x = j["ins"][0]
                if len(j["ins"]) == 1 and x[0] == "a" and x[1] in self.gathered:
                    j["gather"] = (x[1], x[2])
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

        # a mixing fold whose every factor is the output of a sum job that nobody else reads has no backward launch: each of
        # those sum jobs forms its own gradient from the MIXING fold's gradient list (ck_sum_job.mix_out), and one job per slot
        # leaves d w[:, h]; the softmax behind the coefficients is differentiated by one launch for all such folds at the end
        sum_of = {j["out"]: j for j in self.sum_jobs}
        self.mix_fold_bwd = FOLD_MIX_BWD
        for r in (self.mix_jobs if FOLD_MIX_BWD else []):
            ok = True
            for h, sl in enumerate(r["slots"]):
                for x in sl:
                    if x not in sum_of or gsrc.get(x) != [("x", r["gx0"] + h)] or "mix" in sum_of[x]:
                        ok = False
            if not ok or r["S"] > 4:
                continue
            r["folded"] = True
            for h, sl in enumerate(r["slots"]):
                for k, x in enumerate(sl):
                    sum_of[x]["mix"] = {"job": r, "h": h, "partners": [y for y in sl if y is not x], "writer": k == 0}
        for j in self.sum_jobs:
            j["gof"] = j["mix"]["job"]["out"] if "mix" in j else j["out"]  # the block whose gradient list the job reads
        live_mix = [r for r in self.mix_jobs if not r.get("folded")]
        for r in live_mix:
            r["gof"] = r["out"]
        readers: dict[tuple, int] = {}
        wanted = [j["gof"] for j in self.sum_jobs + live_mix] + [("a", i, f) for i in self.inputs for f in range(c.layers[i].num_folds)]
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
        for j in self.sum_jobs + live_mix:
            j["g"] = list(self._sources(j["gof"]))
            if not j["g"]:
                return f"layer {j['layer']} fold {j['fold']} feeds nothing"
        # backward levels: the root is level 0; a job follows the writers of its gradient list
        writer: dict[tuple, dict] = {}
        for r in range(len(order)):
            writer[("x", g0 + r)] = {"lb": 0}
        for j in self.sum_jobs:
            writer[j["gx"]] = j
        for j in live_mix:
            for h in range(j["H"]):
                writer[("x", j["gx0"] + h)] = j
        for j in self.gsum_jobs:
            writer[j["out"]] = j

        def lb(j: dict) -> int:
            if "lb" not in j:
                j["lb"] = 1 + max(lb(writer[gsid]) for gsid in (j["g"] if "g" in j else j["ins"]))
            return j["lb"]

        for j in self.sum_jobs + live_mix + self.gsum_jobs:
            lb(j)
        # the gradient of every input-layer fold, gathered into a contiguous (F, B, 64) block per layer for its backward
        self.input_g: dict[int, dict] = {}
        for i in self.inputs:
            Fi = c.layers[i].num_folds
            first = -1 if (i in self.gauss or i in self.cat) else extra(Fi)
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

    def _uncovered(self) -> list[int]:
        """Input layers whose parameters no job epilogue updates (their gradients go to the flat buffer; the fused step runs
        the optimizer on their tensors' ranges and re-evaluates their parameter graphs at its start)."""
        return [i for i in self.inputs if i not in self.gauss and i not in self.cat]

    def _opt_state(self) -> torch.Tensor:
        """The DEVICE ck_opt_state of the fused optimizer (created on first use; its constants follow the trainer's)."""
        tr = self.tr
        key = (float(tr.lr), tuple(float(b) for b in tr.betas), float(tr.eps), tr.optimizer)
        if self._opt is None:
            o = capi.OptState()
            o.lr, o.b1, o.b2, o.eps, o.bc1, o.bc2 = tr.lr, tr.betas[0], tr.betas[1], tr.eps, 1.0, 1.0
            o.step, o.skipped, o.skip_now, o.kind = 0, 0, 0, 1 if tr.optimizer == "adam" else 0
            o.b1d, o.b2d = float(tr.betas[0]), float(tr.betas[1])  # the bias corrections are formed in double (torch.optim.Adam does)
            self._opt = torch.frombuffer(bytearray(bytes(o)), dtype=torch.uint8).to(self.c.device)
            self._opt_key = key
        elif key != self._opt_key:  # (the learning rate was changed between steps: the first 16 bytes)
            head = torch.tensor([tr.lr, tr.betas[0], tr.betas[1], tr.eps], dtype=torch.float32).view(torch.uint8)
            self._opt[:16].copy_(head.to(self.c.device))
            self._opt[40:56].copy_(torch.tensor([tr.betas[0], tr.betas[1]], dtype=torch.float64).view(torch.uint8).to(self.c.device))
            self._opt_key = key
        return self._opt

    def opt_counters(self) -> tuple[int, int]:
        """(steps taken, steps dropped) of the fused optimizer (a device read)."""
        if self._opt is None:
            return 0, 0
        v = self._opt[24:32].cpu().view(torch.int32)
        return int(v[0]), int(v[1])

    def bind(self, B: int) -> dict:
        tr, c = self.tr, self.c
        bd = c._bind(B)
        st = self._bound.get(B)
        if st is not None and st["arena_ptr"] == bd.arena.data_ptr() and st["store_version"] == c.store.version:
            return st
        if st is not None:
            for pr in st["prog"].values():
                capi.load().ck_program_destroy(pr)
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

        grads, flat_p = tr.grads, tr._flat_param

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
        keep: list[torch.Tensor] = [extra]

        def upload(tab: np.ndarray, both_modes: bool) -> dict:
            """The DEVICE copy of a job table -- for the backward launches one per mode (1: d theta to the flat gradient,
            2: the optimizer's update in the epilogue)."""
            out = {}
            for mode in ((1, 2) if both_modes else (1,)):
                if both_modes:
                    tab["mode"] = mode
                t = torch.from_numpy(tab.view(np.uint8).reshape(len(tab), -1).copy()).to(dev)
                keep.append(t)
                out[mode] = t
            return out

        def splits_for(n_jobs: int, max_split: int) -> int:
            return int(max(1, min(max_split, -(-2 * n_cu // max(1, n_jobs)))))

        def sum_config(n_jobs: int, backward: bool) -> tuple[int, int]:
            """(row splits per job, waves per workgroup) of a sum launch.  The launch runs in ROUNDS of as many workgroups as the
            chip holds (backward: 2 per CU with 4 waves, 1 with 8; forward: 4 per CU) and a round lasts as long as one unit, so
            1040 jobs on 512 slots take three rounds where 2.03 would do -- finer units waste less of the last round, at a fixed
            cost per unit.  Measured [MI355X, scripts/exp_jobs_fixed.py]: a backward unit costs ~10 us beside its tiles (launch,
            weights staged, accumulators reduced, optimizer epilogue), ~10 us more when its partial sums of dW go through memory,
            and a tile ~10 us of a wave that has its SIMD to itself, ~14 us when two waves share it; forward ~3 us + 6 us per
            tile.  Eight waves halve a unit's chain of tiles (what the few-fold levels at the top of a circuit consist of)."""
            best, best_t = (1, 4), None
            for waves in ((4, 8) if backward else (4,)):
                slots = n_cu * ((2 if waves == 4 else 1) if backward else 4)
                for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                    if sp > 1 and -(-tiles // sp) < waves:  # (at least a tile per wave)
                        break
                    per_wave = -(-(-(-tiles // sp)) // waves)
                    if backward:
                        shared = waves == 8 or n_jobs * sp > n_cu
                        unit = 10.0 + (10.0 if sp > 1 else 0.0) + (14.0 if shared else 10.0) * per_wave
                    else:
                        unit = 3.0 + 6.0 * per_wave
                    t = -(-n_jobs * sp // slots) * unit
                    if best_t is None or t < best_t - 1e-9:
                        best, best_t = (sp, waves), t
            return best

        folded = [r for r in self.mix_jobs if r.get("folded")]
        mix_dw = torch.zeros(max(1, sum(K * r["H"] for r in folded)), dtype=torch.float32, device=dev)
        keep.append(mix_dw)
        mix_dw_off: dict[int, int] = {}
        off_f = 0
        for r in folded:
            mix_dw_off[id(r)] = off_f
            off_f += K * r["H"]

        def mix_dw_ptr(r: dict) -> int:
            return mix_dw.data_ptr() + 4 * mix_dw_off[id(r)]

        def sum_layout(n_jobs: int, backward: bool) -> tuple[list[int], int]:
            """(row splits of every job, waves per workgroup): `sum_config`'s uniform answer, or -- a backward launch of more
            jobs than the chip holds -- whole jobs for the full rounds and only the REMAINDER cut fine, issued last: 1060 jobs
            on 512 slots are two rounds of whole jobs + 36 jobs in 8 pieces each instead of three rounds."""
            sp, waves = sum_config(n_jobs, backward)
            if not backward:
                return [sp] * n_jobs, waves

            def unit(s_: int, w_: int, shared: bool) -> float:
                per_wave = -(-(-(-tiles // s_)) // w_)
                return 10.0 + (10.0 if s_ > 1 else 0.0) + (14.0 if shared else 10.0) * per_wave

            slots_u = n_cu * (2 if waves == 4 else 1)
            t_uniform = -(-n_jobs * sp // slots_u) * unit(sp, waves, waves == 8 or n_jobs * sp > n_cu)
            best = ([sp] * n_jobs, waves, t_uniform)
            for w_ in (4, 8):
                slots = n_cu * (2 if w_ == 4 else 1)
                full = (n_jobs // slots) * slots
                rem = n_jobs - full
                if full == 0 or rem == 0:
                    continue
                for sr in (2, 3, 4, 6, 8, 12, 16):
                    if -(-tiles // sr) < w_:
                        break
                    t = (full // slots) * unit(1, w_, True) + -(-rem * sr // slots) * unit(sr, w_, True)
                    if t < best[2] - 1e-9:
                        best = ([1] * full + [sr] * rem, w_, t)
            return best[0], best[1]

        def sum_table(jobs: list[dict], backward: bool) -> tuple[dict, int, int]:
            splits, waves = sum_layout(len(jobs), backward)
            rows_of, first_unit, n_units = [], [], 0
            for ns_j in splits:  # (a job's pieces: whole 32-row tiles, as equal as they come)
                rp = -(-tiles // ns_j) * 32
                rows_of.append(rp)
                first_unit.append(n_units)
                n_units += -(-B // rp)
            tab = np.zeros(n_units, dtype=np.dtype(capi.SUM_JOB_DTYPE))
            part = tick = None
            if backward and n_units > len(jobs):
                part = torch.zeros(n_units * 4096, dtype=torch.float32, device=dev)
                tick = torch.zeros(len(jobs), dtype=torch.int32, device=dev)
                keep.extend([part, tick])
            for n, j in enumerate(jobs):
                rows_per = rows_of[n]
                ns = -(-B // rows_per)
                u0 = first_unit[n]
                xrow = Cg = 0
                if "gather" in j:  # the input is a Categorical fold: the pool entry is its table, rows picked by the batch column
                    gi, gf = j["gather"]
                    gl = c.layers[gi]
                    Cg = gl.num_categories
                    xrow = bd.xt_i.data_ptr() + int(gl.scope_idx[gf, 0]) * B * 4
                    ioff, inum = len(pool), 1
                    pool.append(gl._table.data_ptr() + gf * (Cg + 1) * K * 4)
                else:
                    ioff, inum = put(j["ins"])
                goff, gnum = put(j["g"]) if backward else (0, 0)
                th, m1, m2 = theta_ptrs(j["theta"])
                w = self._weight_ptr(j["w"])
                mx = j.get("mix") if backward else None
                if mx is not None:
                    mj = mx["job"]
                    poff, pnum = put(mx["partners"])
                    m_out, m_w = addr(mj["out"]), self._weight_ptr(mj["w"])
                    m_dw = mix_dw_ptr(mj) if mx["writer"] else 0
                for sp in range(ns):
                    r = tab[u0 + sp]
                    r["xrow"], r["C"] = xrow, Cg
                    if mx is not None:
                        r["mix_out"], r["mix_w"], r["mix_dw"] = m_out, m_w, m_dw
                        r["partner_off"], r["n_partner"], r["mix_h"], r["mix_H"] = poff, pnum, mx["h"], mj["H"]
                    r["w"], r["out"], r["gx"], r["dtheta"] = w, addr(j["out"]), addr(j["gx"]), grad_ptr(j["theta"])
                    r["theta"], r["m1"], r["m2"], r["w_out"] = th, m1, m2, w
                    r["in_off"], r["n_in"], r["g_off"], r["n_g"] = ioff, inum, goff, gnum
                    r["row0"], r["row1"] = sp * rows_per, min(B, (sp + 1) * rows_per)
                    r["split"], r["n_split"], r["mode"] = sp, ns, 1
                    if part is not None and ns > 1:
                        r["part"], r["ticket"] = part.data_ptr() + u0 * 4096 * 4, tick.data_ptr() + n * 4
            return upload(tab, backward), len(tab), waves

        def mix_table(jobs: list[dict], backward: bool) -> tuple[dict, int, int]:
            hmax = max(j["H"] for j in jobs)
            hpad = 2 if hmax <= 2 else 4 if hmax <= 4 else 8 if hmax <= 8 else 16
            ns = splits_for(len(jobs), max(1, B // 64))
            if not backward:
                # the forward has no sums over the rows: eight workgroups per CU instead of two (a workgroup is a chain of one
                # round trip per 16 rows, and 256 threads with 4 KB of LDS leave room for eight of them)
                ns = int(max(1, min(max(1, B // 64), -(-int(os.environ.get("CK_MIX_FWD_WG", "8")) * n_cu // max(1, len(jobs))))))
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
            return upload(tab, backward), len(tab), hmax

        def nsum_table(items: list[tuple[list, int]]) -> tuple[dict, int]:
            tab = np.zeros(len(items), dtype=np.dtype(capi.NSUM_JOB_DTYPE))
            for r, (ins, out) in zip(tab, items):
                r["in_off"], r["n_in"] = put(ins)
                r["out"] = out
            return upload(tab, False), len(tab)

        def gauss_table(i: int) -> tuple[dict, int]:
            l = c.layers[i]
            mean, stddev, _ = l._vals
            recs = self.gauss[i]
            tab = np.zeros(len(recs), dtype=np.dtype(capi.GAUSS_JOB_DTYPE))
            for f, (r, rec, lst) in enumerate(zip(tab, recs, self.input_g[i]["lists"])):
                thm, m1m, m2m = theta_ptrs(rec["mean"])
                ths, m1s, m2s = theta_ptrs(rec["sd"])
                mp, sp_ = mean.data_ptr() + f * K * 4, stddev.data_ptr() + f * K * 4
                r["mean"], r["stddev"] = mp, sp_
                r["x"] = bd.xt.data_ptr() + int(l.scope_idx[f, 0]) * B * 4
                r["dmean"], r["dsd"] = grad_ptr(rec["mean"]), grad_ptr(rec["sd"])
                r["th_mean"], r["m1_mean"], r["m2_mean"] = thm, m1m, m2m
                r["th_sd"], r["m1_sd"], r["m2_sd"] = ths, m1s, m2s
                r["mean_out"] = 0 if mp == thm else mp
                r["sd_out"] = sp_
                r["g_off"], r["n_g"] = put(lst)
                r["vmin"], r["vmax"], r["has_ss"], r["mode"] = rec["vmin"], rec["vmax"], 1 if rec["ss"] else 0, 1
            return upload(tab, True), len(tab)

        def cat_table(i: int) -> tuple[dict, int]:
            l = c.layers[i]
            name = l.probs.graph.nodes[0].config["tensor"]
            Cn = l.num_categories
            tab = np.zeros(l.num_folds, dtype=np.dtype(capi.CAT_JOB_DTYPE))
            for f, (r, lst) in enumerate(zip(tab, self.input_g[i]["lists"])):
                th, m1, m2 = theta_ptrs((name, f))
                tb = l._table.data_ptr() + f * (Cn + 1) * K * 4
                r["x"] = bd.xt_i.data_ptr() + int(l.scope_idx[f, 0]) * B * 4
                r["theta"], r["table"], r["dtheta"] = th, tb, grad_ptr((name, f))
                r["theta_out"], r["m1"], r["m2"], r["table_out"] = th, m1, m2, tb
                r["g_off"], r["n_g"] = put(lst)
                r["mode"] = 1
            return upload(tab, True), len(tab)

        def by_level(jobs: list[dict], key: str) -> dict[int, list[dict]]:
            out: dict[int, list[dict]] = {}
            for j in jobs:
                out.setdefault(j[key], []).append(j)
            return out

        launches: list[tuple] = []  # (what, table(s), n, ...), in issue order
        fs, fm, fn = by_level(self.sum_jobs, "lf"), by_level(self.mix_jobs, "lf"), by_level(self.nsum_jobs, "lf")
        for lv in sorted(set(fs) | set(fm) | set(fn)):
            if lv in fn:
                launches.append(("nsum",) + nsum_table([(j["ins"], addr(j["out"])) for j in fn[lv]]))
            if lv in fs:
                launches.append(("sum_fwd",) + sum_table(fs[lv], False))
            if lv in fm:
                launches.append(("mix_fwd",) + mix_table(fm[lv], False))
        launches.append(("root",))
        bs, bm, bg = (by_level(self.sum_jobs, "lb"), by_level([r for r in self.mix_jobs if not r.get("folded")], "lb"),
                      by_level(self.gsum_jobs, "lb"))
        bi: dict[int, list[int]] = {}
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
                if i in self.gauss:
                    launches.append(("gauss_bwd", i) + gauss_table(i))
                elif i in self.cat:
                    launches.append(("cat_bwd", i) + cat_table(i))
                else:
                    launches.append(("input_bwd", i) + nsum_table([(lst, x0 + (ig["first"] + f) * blk * 4) for f, lst in enumerate(ig["lists"])]))
        if folded:  # the coefficients of the mixing folds without a backward launch: one launch for all of them
            tab = np.zeros(len(folded), dtype=np.dtype(capi.MIX_JOB_DTYPE))
            for r, j in zip(tab, folded):
                th, m1, m2 = theta_ptrs(j["theta"])
                w = self._weight_ptr(j["w"])
                r["w"], r["dtheta"], r["theta"], r["m1"], r["m2"], r["w_out"] = w, grad_ptr(j["theta"]), th, m1, m2, w
                r["part"], r["H"], r["mode"] = mix_dw_ptr(j), j["H"], 1
            launches.append(("mix_params",) + (upload(tab, True), len(tab)))
        # the root launch
        root = self.root
        R = len(root["folds"])
        rin = np.zeros((2, R), dtype=np.int32)
        ptrs = np.zeros((6, R), dtype=np.uint64)  # w, dtheta, theta, m1, m2, w_out
        S_root = max(len(sc["ins"]) for sc in root["folds"])
        for r, sc in enumerate(root["folds"]):  # (lists of one length: shorter ones are padded with the block of zeros)
            rin[0, r], rin[1, r] = put(list(sc["ins"]) + [("x", root["zero"])] * (S_root - len(sc["ins"])))
            l = c.layers[sc["layer"]]
            w = l._w.data_ptr() + sc["fold"] * K * 4
            th, m1, m2 = theta_ptrs(sc["theta"])
            ptrs[:, r] = (w, grad_ptr(sc["theta"]), th, m1, m2, w)
        rin_d = torch.from_numpy(rin).to(dev)
        ptrs_d = torch.from_numpy(ptrs.view(np.int64)).to(dev)
        n_wg = int(max(1, min(64, B // 4)))  # (a row per wave up to 256 rows: the launch is a chain of row-long round trips)
        rpart = torch.zeros(n_wg * 1042, dtype=torch.float32, device=dev)
        rtick = torch.zeros(1, dtype=torch.int32, device=dev)
        seed = torch.zeros(B, dtype=torch.float32, device=dev)
        pool_d = torch.from_numpy(np.asarray(pool, dtype=np.uint64).view(np.int64)).to(dev)
        keep.extend([rin_d, ptrs_d, rpart, rtick, seed, pool_d])
        opt = self._opt_state()

        def root_args(mode: int):
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
            validate = c.validate_inputs and c._int_input
            # mode 1: the circuit's own flag (latched after the step by the trainer); mode 2: `ck_opt_tick` has moved it into
            # the optimizer state's skip_now (byte 32) by the time the root launch runs
            ra.opt = opt.data_ptr() if mode == 2 else None
            ra.bad_flag = (opt.data_ptr() + 32) if mode == 2 else (c._bad_input.data_ptr() if validate else None)
            ra.seed_const, ra.R, ra.B, ra.mode, ra.n_wg, ra.S = 0.0, R, B, mode, n_wg, S_root
            return ra

        st = {"arena_ptr": bd.arena.data_ptr(), "store_version": c.store.version, "keep": keep, "launches": launches,
              "root": {1: root_args(1), 2: root_args(2)}, "pool": pool_d, "seed": seed, "seed_value": None, "extra": extra, "x0": x0,
              "prog": {}, "dT": {}}
        while len(self._bound) >= 4:
            old = self._bound.pop(next(iter(self._bound)))
            for pr in old["prog"].values():
                capi.load().ck_program_destroy(pr)
        self._bound[B] = st
        return st

    # ---- the launch list ----------------------------------------------------------------------------------------------------
    def _program(self, st: dict, B: int, mode: int):
        prog = st["prog"].get(mode)
        if prog is None:
            bd = self.c._bind(B)
            prog = C.c_void_p()
            capi.call("ck_program_begin", C.byref(prog))
            try:
                self._enqueue(bd, st, B, mode, 0)
            finally:
                capi.call("ck_program_end", prog)
            st["prog"][mode] = prog
        return prog

    def _enqueue(self, bd, st: dict, B: int, mode: int, stream: int) -> None:
        """mode 1: parameters, forward, backward with d theta into the trainer's flat gradient (the optimizer launch and the
        collective follow outside).  mode 2 (one rank): the optimizer runs in the job epilogues -- `ck_opt_tick` first, the
        parameter graphs of the layers no epilogue covers re-evaluated, and the optimizer on their tensors at the end."""
        tr, c = self.tr, self.c
        pool = st["pool"].data_ptr()
        blk = B * K
        opt = self._opt_state().data_ptr() if mode == 2 else None
        validate = c.validate_inputs and c._int_input
        if mode == 2:
            capi.call("ck_opt_tick", opt, c._bad_input.data_ptr() if validate else None, tr._bad_seen.data_ptr() if validate else None, stream)
            for i in self._uncovered():  # their parameter graphs, as every forward of the reference evaluates them
                idx = c._jobs_of_layer.get(i)
                if idx:
                    c._batch.subset(idx).launch(stream)
                else:
                    c.layers[i].prepare(stream, batched=False)
        else:
            c._enqueue_params(stream)  # every parameter graph, once per step (parameters/parameter.py:180-188)
        D = c.plan.num_variables
        for i in self.inputs:
            if i in self.gathered:
                continue  # (its consumers read the table)
            l = c.layers[i]
            l.launch_input(bd.xt if l.wants_float_input else bd.xt_i, D, bd.views[i], B, stream)
        for la in st["launches"]:
            what = la[0]
            if what == "nsum":
                capi.call("ck_jobs_nsum", la[1][1].data_ptr(), la[2], pool, blk, stream)
            elif what == "sum_fwd":
                capi.call("ck_jobs_sum64_fwd", la[1][1].data_ptr(), la[2], pool, stream)
            elif what == "mix_fwd":
                capi.call("ck_jobs_mix_fwd", la[1][1].data_ptr(), la[2], pool, la[3], stream)
            elif what == "root":
                capi.call("ck_jobs_root", C.byref(st["root"][mode]), stream)
            elif what == "sum_bwd":
                capi.call("ck_jobs_sum64_bwd", la[1][mode].data_ptr(), la[2], pool, opt, la[3], stream)
            elif what == "mix_bwd":
                capi.call("ck_jobs_mix_bwd", la[1][mode].data_ptr(), la[2], pool, la[3], blk, opt, stream)
            elif what == "mix_params":
                capi.call("ck_jobs_mix_params", la[1][mode].data_ptr(), la[2], opt, stream)
            elif what == "gauss_bwd":
                capi.call("ck_jobs_gauss_bwd", la[2][mode].data_ptr(), la[3], pool, B, opt, stream)
            elif what == "cat_bwd":
                capi.call("ck_jobs_cat_bwd", la[2][mode].data_ptr(), la[3], pool, B, c.layers[la[1]].num_categories, opt, stream)
            elif what == "input_bwd":
                i = la[1]
                capi.call("ck_jobs_nsum", la[2][1].data_ptr(), la[3], pool, blk, stream)
                self._input_backward(i, bd, st, B, stream)
        if mode == 2:  # the tensors of the layers no epilogue covers: the optimizer on their ranges of the flat buffers
            for i in self._uncovered():
                for name in self._tensors_of(i):
                    t, g = c.store[name], tr.grads[name]
                    off = t.data_ptr() - tr._flat_param.data_ptr()
                    capi.call("ck_opt_step_range", t.data_ptr(), g.data_ptr(), None, (tr._m1.data_ptr() + off) if tr._m1 is not None else None,
                              (tr._m2.data_ptr() + off) if tr._m2 is not None else None, t.numel(), opt, stream)

    def _tensors_of(self, i: int) -> list[str]:
        l = self.c.layers[i]
        names: list[str] = []
        for p in l.params.values():
            for n in p.graph.nodes:
                if n.op == "tensor" and n.config["tensor"] not in names:
                    names.append(n.config["tensor"])
        return names

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
        else:  # a Gaussian layer with parameter graphs the job epilogue does not know
            mean, stddev, _ = l._vals
            dm = st["dT"].get((i, "m"))
            if dm is None:
                dm = st["dT"][(i, "m")] = torch.zeros_like(mean)
                st["dT"][(i, "s")] = torch.zeros_like(stddev)
            ds = st["dT"][(i, "s")]
            capi.call("ck_gaussian_bwd", g, bd.xt.data_ptr(), l._scope(dev).data_ptr(), mean.data_ptr(), stddev.data_ptr(), dm.data_ptr(),
                      ds.data_ptr(), l.num_folds, B, K, stream)
            for name in self._tensors_of(i):  # (the parameter backward ADDS into the tensors' gradients)
                t = tr.grads[name]
                capi.call("ck_fill_f32", t.data_ptr(), t.numel(), 0.0, stream)
            l.mean.backward(dm, tr.grads, stream)
            l.stddev.backward(ds, tr.grads, stream)

    # ---- one step -------------------------------------------------------------------------------------------------------------
    def _launch(self, x: torch.Tensor, gB: float, mode: int) -> torch.Tensor:
        c = self.c
        B = int(x.shape[0])
        st = self.bind(B)
        bd = c._bind(B)
        with torch.cuda.device(c.device):
            stream = torch.cuda.current_stream(c.device).cuda_stream
            if mode == 2:
                self._opt_state()
                if self._own_state != c.store.state():
                    # somebody else changed a parameter since this object last derived them: all graphs, once
                    c._enqueue_params(stream)
            prog = self._program(st, B, mode)
            if st["seed_value"] != -1.0 / gB:
                capi.call("ck_fill_f32", st["seed"].data_ptr(), B, -1.0 / gB, stream)
                st["seed_value"] = -1.0 / gB
            xf, xi = c._prepare_input(x)
            c._stage_input(bd, xf, xi, stream)
            capi.call("ck_program_launch", prog, 0, stream)
            if mode == 2:
                c.store.touch()
                self._own_state = c.store.state()
        return bd.ll

    def loss_and_grads(self, x: torch.Tensor, gB: float) -> torch.Tensor:
        """Forward + backward of ``-(1 / gB) sum_b log p(x_b)`` over the recorded launch list; the parameter gradients land in
        the trainer's flat buffer; returns the circuit's [sum log p, rows] pair (device, overwritten by the next call at this
        batch size)."""
        return self._launch(x, gB, 1)

    def step(self, x: torch.Tensor, gB: float) -> torch.Tensor:
        """One optimisation step with the optimizer inside the job epilogues (a single rank: no gradient leaves the launch
        that computed it, so there is nothing a collective could reduce)."""
        return self._launch(x, gB, 2)

    def num_launches(self, B: int, mode: int = 2) -> int:
        st = self.bind(B)
        return int(capi.load().ck_program_num_ops(self._program(st, B, mode)))
