#!/usr/bin/env python3
"""Headline benchmark: log-likelihood evaluations / second of the folded log-space forward.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]; configs[2] at N=8): 784-variable QuadTree-2 circuit,
Categorical-256 inputs, CP sum layers, K=32, fp32, batch 4096 PER GPU (weak scaling), synthetic
int64 batches resident in HBM, closed-form random-init parameters (cirkit_amd.initializers).
One "step" = one forward of the rank's 4096-row batch (parameter softmax/log recomputed every step,
as the reference does) + the device-side sum of the log-likelihoods (+ for N > 1 the single RCCL
all-reduce of the [sum, count] pair).

Prints ONE JSON line on rank 0 (contract in the task description), with
  roofline     -- dominant kernel: algorithmic flops / bytes (SURVEY.md section 8 d) per launch / HIP-event duration
                  against the roofline that binds it (fp32 MFMA 157.3 TFLOP/s for the fused leaf kernel, whose HBM
                  traffic is a fraction of the algorithmic bytes; 8 TB/s HBM3E otherwise), the other view beside it,
                  the PMC-measured HBM bytes per launch (`traffic`), and the whole-forward figure;
  cpu_baseline -- the CPU oracle (op-for-op port of the reference's torch-CPU path) timed on the
                  host cores of this box on a bounded sample (rank 0, N = 1 only).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured copy
FP32_MFMA_PEAK_TF = 157.3  # dense fp32-input MFMA peak (= fp32 vector peak), same guide


def other_configs(device, stream, B: int) -> dict:
    """BASELINE configs 4 and 5 (parity-test cases, tests/test_gpu_parity.py) timed for reference:
    plans built natively, closed-form parameters, synthetic batch, recorded launch list replayed per call, HIP events on
    the launch stream.  Informational only -- `value` and `roofline` are config 2."""
    import numpy as np
    import torch

    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.functional import squared_partition_plan
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    def time_forward(hc, x, steps=20):
        with torch.cuda.stream(stream):
            for _ in range(3):
                hc(x)
            torch.cuda.synchronize(device)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            for a, b in ev:
                a.record(stream)
                hc(x)
                b.record(stream)
            torch.cuda.synchronize(device)
        return float(np.median([a.elapsed_time(b) for a, b in ev]))

    out = {}
    g = torch.Generator().manual_seed(4)
    plan4 = image_data((1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64,
                       sum_product_layer="cp", num_sum_units=64)
    hc = HipCircuit(plan4, init_plan_tensors(plan4), device=device)
    ms = time_forward(hc, torch.randn((B, 784), generator=g).to(device))
    alg = plan4.algorithmic_bytes(B)["total"]
    out["config4"] = {
        "workload": f"Poon-Domingos 28x28 (delta 4), Gaussian leaves, CP sum layers, mixing layers, K=64, batch {B}, "
                    "38 folded layers; CP blocks + mixing layers fused per region (cirkit_amd/csrc/ck_cp.hip)",
        "ms_per_forward": ms, "evals_per_s": B / ms * 1e3, "algorithmic_bytes": alg,
        "hbm_roofline_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }
    del hc
    plan5 = image_data((1, 28, 28), "quad-tree-2", input_layer="embedding", num_input_units=32, sum_product_layer="cp-t",
                       num_sum_units=32, sum_weight_activation="none", semiring="complex-lse-sum")
    t5 = init_plan_tensors(plan5)
    hc = HipCircuit(plan5, t5, device=device)
    ms = time_forward(hc, torch.randint(0, 256, (B, 784), generator=g).to(device))
    alg = plan5.algorithmic_bytes(B)["total"]
    hz = HipCircuit(squared_partition_plan(plan5), hc.store, device=device)
    ms_z = time_forward(hz, None)
    out["config5"] = {
        "workload": f"squared (SoS) circuit: QuadTree-2 28x28, Embedding-256, CP-T, K=32, complex-lse-sum, batch {B}; "
                    "Z = integral |c|^2 built from the plan of c (cirkit_amd/functional.py)",
        "ms_per_forward": ms, "evals_per_s": B / ms * 1e3, "algorithmic_bytes": alg,
        "hbm_roofline_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "ms_partition_function": ms_z,
    }
    del hc, hz
    # The only forward timing the reference publishes (notebooks/compilation-options.ipynb:594):
    # QuadGraph 28x28, Categorical-256, Tucker layers, K = 64, batch 128, fold + optimize: 38.6 ms on an unnamed
    # CUDA GPU.  Different hardware and not the north-star metric, so it stays out of `vs_baseline`.
    plan_nb = image_data((1, 28, 28), "quad-graph", input_layer="categorical", num_input_units=64,
                         sum_product_layer="tucker", num_sum_units=64)
    hc = HipCircuit(plan_nb, init_plan_tensors(plan_nb), device=device)
    ms = time_forward(hc, torch.randint(0, 256, (128, 784), generator=g).to(device))
    out["notebook_quadgraph_tucker_k64_b128"] = {
        "workload": "QuadGraph 28x28, Categorical-256, Tucker, K=64, batch 128 (the reference's compilation-options "
                    "notebook); Tucker layers on MFMA (cirkit_amd/csrc/ck_gemm.hip)",
        "ms_per_forward": ms, "evals_per_s": 128 / ms * 1e3,
        "reference_published_ms": 38.6, "reference_hardware": "unnamed CUDA GPU (notebook output)",
    }
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="rows per GPU (default: the BASELINE config)")
    ap.add_argument("--no-graph", action="store_true", help="never replay as a hipGraph (only lists of more than 64 launches are by default)")
    ap.add_argument("--fuse", type=int, default=-1, help="-1: full leaf fusion (default), 0: layer-wise, n: n CP-T levels")
    ap.add_argument("--contraction", default="f32", choices=["f32", "f16x3"],
                    help="K=32 sum layers: exact fp32 MFMA (default) or 3-term split-fp16 MFMA with fp32 accumulation")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary f16x3 measurement")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short measurements of BASELINE configs 4 and 5 (never part of `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-breakdown", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(
                f"--gpus {args.gpus} needs one process per GPU: launch with "
                f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ..."
            )
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    # one process per GPU; BENCH_DIST_BACKEND=gloo lets the N > 1 path be exercised on a 1-GPU box
    # (ranks then share the device; RCCL itself refuses two ranks on one GPU)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    device = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # BASELINE configs[1], built natively (cirkit_amd/templates.py; identical to the plan the reference
    # compiles -- tests/test_templates.py pins it against the committed reference fixture)
    plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32,
                      sum_product_layer="cp", num_sum_units=32)
    tensors = init_plan_tensors(plan)
    B = args.batch
    fuse = True if args.fuse < 0 else (False if args.fuse == 0 else args.fuse)
    circuit = HipCircuit(plan, tensors, device=device, use_graph=not args.no_graph, fuse=fuse,
                         contraction=args.contraction)
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randint(0, 256, (B, plan.num_variables), generator=g).to(device)  # int64, like the reference

    stream = torch.cuda.Stream(device)

    def timed_region(circ, steps, warmup):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides; returns
        (wall seconds, mean HIP-event ms per step on the launch stream, final [sum, count])."""
        last = [None]
        # N > 1: the 16-byte all-reduce of step k runs on RCCL's stream WHILE step k + 1 computes (its
        # input is copied out of the circuit's [sum, count] buffer, which the next step overwrites);
        # a ring of buffers, each reused only after its previous collective has completed
        ring = [torch.zeros(2, dtype=torch.float64, device=device) for _ in range(8)] if world > 1 else []
        works: list = [None] * len(ring)
        count = [0]

        def step() -> None:
            ll = circ.log_likelihood_sum(x)  # forward + device-side sum, enqueued on `stream`
            if world > 1:
                i = count[0] % len(ring)
                count[0] += 1
                if works[i] is not None:
                    works[i].wait()  # stream-level wait on a collective issued 8 steps ago
                ring[i].copy_(ll)
                works[i] = dist.all_reduce(ring[i], op=dist.ReduceOp.SUM, async_op=True)  # the ONE exchange: 16 bytes over xGMI
                ll = ring[i]
            last[0] = ll

        def drain() -> None:
            for w in works:
                if w is not None:
                    w.wait()

        with torch.cuda.stream(stream):
            for _ in range(warmup):
                step()
            drain()
            torch.cuda.synchronize(device)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(device)
            # ONE event pair around the K steps (an event record between steps costs a command-processor
            # round trip of several microseconds, comparable to a whole kernel of this forward)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(steps):
                step()
            drain()  # every collective of the timed steps has completed before the clock stops
            e1.record(stream)
            torch.cuda.synchronize(device)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(device)
            wall = time.perf_counter() - t0
        pair = last[0].cpu()
        return wall, e0.elapsed_time(e1) / steps, pair

    elapsed, step_ms_events, pair = timed_region(circuit, args.steps, args.warmup)
    total_nll = float(pair[0])
    total_rows = float(pair[1])

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    alg = plan.algorithmic_bytes(B)  # 615.7 KB per evaluation at this config

    result = {
        "metric": "log-likelihood evals/sec (batch 4096 per GPU) on 784-var QuadTree PC",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "QuadTree-2 28x28 (784 vars), Categorical-256 leaves, CP sum layers, K=32, "
                        f"batch {B}/GPU, lse-sum, fold+optimize plan (12 folded layers)",
            "global_batch": world * B,
            "parallelism": f"dp{world} (batch-sharded, replicated parameters, one all-reduce of the summed LL)",
            # launch lists are replayed eagerly by the native executor; a hipGraph only beyond 64 launches (cirkit_amd/circuit.py)
            "hip_graph": circuit.replays_as_graph(B, with_ll=True),
            "fused_leaf_levels": [g.depth for g in circuit._groups],
            "fused_tail_layers": len(circuit._tail),
            "contraction": args.contraction,
            "dense_on_table": circuit.dense_on_table,
            "params_recomputed_every_step": True,
        },
        "check": {"mean_ll": total_nll / max(total_rows, 1.0), "rows": total_rows},
    }

    # Secondary figure (never `value`): the same step with the split-fp16 contraction.
    variants = {}
    if args.contraction == "f32" and not args.no_variants and world == 1:  # single-GPU extras only
        alt = HipCircuit(plan, tensors, device=device, use_graph=not args.no_graph, fuse=fuse, contraction="f16x3")
        w2, ms2, pair2 = timed_region(alt, args.steps, args.warmup)
        if world > 1:
            t2 = torch.tensor([w2], dtype=torch.float64, device=device)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            w2 = float(t2.item())
        variants["contraction=f16x3"] = {
            "what": "K=32 sum layers contract with 3-term split-fp16 MFMA products, fp32 accumulation "
                    "(~22-bit significand; cirkit_amd/csrc/ck_tile.h); everything else identical",
            "value": world * B * args.steps / w2,
            "ms_per_step": 1e3 * w2 / args.steps,
            "mean_ll": float(pair2[0]) / max(float(pair2[1]), 1.0),
            "mean_ll_rel_diff_vs_f32": abs(float(pair2[0]) - total_nll) / abs(total_nll),
        }
        del alt
        # "... and additionally reported cached": derived parameters (softmax, log
        # tables, tiled weights) kept from the previous step -- the serving configuration.
        alt = HipCircuit(plan, tensors, device=device, use_graph=not args.no_graph, fuse=fuse, cache_params=True)
        w3, ms3, pair3 = timed_region(alt, args.steps, args.warmup)
        if world > 1:
            t3 = torch.tensor([w3], dtype=torch.float64, device=device)
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
            w3 = float(t3.item())
        # Two forwards in flight (cirkit_amd.circuit.HipCircuitStreams): the small latency-bound kernels of
        # one step (parameter prologue, fused tail) fill the bubbles of the other step's leaf kernel.
        from cirkit_amd.circuit import HipCircuitStreams

        pool = HipCircuitStreams(plan, circuit.store, n=2, device=device, wait_for_input=False,
                                 use_graph=not args.no_graph, fuse=fuse)  # x is resident
        with torch.cuda.stream(stream):
            for _ in range(max(args.warmup, 4)):
                pool.log_likelihood_sum(x)
        pool.synchronize()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(args.steps):
                ll, st = pool.log_likelihood_sum(x)
                if world > 1:
                    with torch.cuda.stream(st):
                        dist.all_reduce(ll, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        w4 = time.perf_counter() - t0
        if world > 1:
            t4 = torch.tensor([w4], dtype=torch.float64, device=device)
            dist.all_reduce(t4, op=dist.ReduceOp.MAX)
            w4 = float(t4.item())
        del pool
        variants["streams=2"] = {
            "what": "steps issued alternately on two HIP streams (HipCircuitStreams: two circuits sharing the raw "
                    "parameters): consecutive forwards overlap on the device; per-step latency is unchanged",
            "value": world * B * args.steps / w4,
            "ms_per_step": 1e3 * w4 / args.steps,
        }
        variants["cache_params=True"] = {
            "what": "parameter graphs evaluated once and reused while the parameters do not change "
                    "(the reference, and `value`, recompute them inside every step)",
            "value": world * B * args.steps / w3,
            "ms_per_step": 1e3 * w3 / args.steps,
            "mean_ll_rel_diff_vs_default": abs(float(pair3[0]) - total_nll) / abs(total_nll),
        }
        del alt

    if rank == 0:
        fwd_ms = step_ms_events
        roof = {
            "bound": "hbm",
            "unit": "GB/s",
            "peak": HBM_PEAK_GBS,
            "traffic": None,  # filled below from the committed rocprofv3 PMC passes, if they match this config
            "forward": {
                "algorithmic_bytes": alg["total"],
                "avg_ms": fwd_ms,
                "achieved": alg["total"] / (fwd_ms * 1e-3) / 1e9,
                "frac": alg["total"] / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
        }
        if not args.no_kernel_breakdown:
            with torch.cuda.stream(stream):
                rows = circuit.profile_kernels(x, iters=10)
            # aggregate per kernel
            agg: dict[str, dict] = {}
            for r in rows:
                a = agg.setdefault(r["kernel"], {"ms": 0.0, "bytes": 0.0, "flops": 0.0, "launches": 0})
                a["ms"] += r["ms"]
                a["bytes"] += r["algorithmic_bytes"]
                a["flops"] += r.get("algorithmic_flops", 0.0)
                a["launches"] += 1
            dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
            name, a = dom
            hbm_view = {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": a["bytes"] / a["launches"],
                "achieved": a["bytes"] / (a["ms"] * 1e-3) / 1e9,
                "frac": a["bytes"] / (a["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            }
            # the same kernel against the matrix roofline of its dtype (fp32-input MFMA, 157.3 TFLOP/s dense,
            # MI355X_MICROARCH.md)
            mfma_view = {
                "bound": "mfma", "unit": "TFLOP/s", "peak": FP32_MFMA_PEAK_TF,
                "algorithmic_flops_per_launch": a["flops"] / a["launches"],
                "achieved": a["flops"] / (a["ms"] * 1e-3) / 1e12,
                "frac": a["flops"] / (a["ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
            }
            # Which roofline binds: a cross-layer-fused kernel moves a fraction of the algorithmic bytes of the layers
            # it covers (PMC: `traffic`), so its algorithmic-bytes rate can exceed the HBM peak -- it is then bound by
            # fp32 MFMA issue and that is the fraction reported; the other view is kept beside it.
            primary, other = (mfma_view, hbm_view) if hbm_view["frac"] > 1.0 and a["flops"] > 0 else (hbm_view, mfma_view)
            roof.update(
                {
                    "bound": primary["bound"], "unit": primary["unit"], "peak": primary["peak"],
                    "kernel": name,
                    "launches_per_step": a["launches"],
                    "avg_us_per_launch": 1e3 * a["ms"] / a["launches"],
                    "algorithmic_bytes_per_launch": a["bytes"] / a["launches"],
                    "algorithmic_flops_per_launch": a["flops"] / a["launches"],
                    "achieved": primary["achieved"],
                    "frac": primary["frac"],
                    ("hbm_view" if other is hbm_view else "mfma_view"): other,
                    "kernels": {
                        k: {
                            "launches": v["launches"],
                            "ms_per_step": v["ms"],
                            "algorithmic_GB_per_s": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None,
                        }
                        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
                    },
                }
            )
        else:
            roof.update({"kernel": "forward program", "achieved": roof["forward"]["achieved"], "frac": roof["forward"]["frac"]})
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            with open(tj, encoding="utf-8") as f:
                tr = json.load(f)
            key = f"fuse={[g.depth for g in circuit._groups]},tail={len(circuit._tail)},contraction={args.contraction},B={B}"
            if key in tr:
                roof["traffic"] = tr[key].get(roof.get("kernel", ""), {}).get("hbm_bytes_per_launch")
                roof["traffic_detail"] = tr[key]
        result["roofline"] = roof

        if world == 1 and not args.no_cpu_baseline:
            from oracle.torch_oracle import as_torch, evaluate_plan

            tt = as_torch(tensors)
            xs = x[:B].cpu()
            # pick the host thread count that serves this op mix best (ATen's small-op overheads
            # make "all cores" far from optimal), on a 512-row probe
            best_thr, best_rate = 1, 0.0
            ncpu = os.cpu_count() or 1
            for thr in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
                torch.set_num_threads(thr)
                evaluate_plan(plan, tt, xs[:512])
                t1 = time.perf_counter()
                evaluate_plan(plan, tt, xs[:512])
                rate = 512 / (time.perf_counter() - t1)
                if rate > best_rate:
                    best_thr, best_rate = thr, rate
            torch.set_num_threads(best_thr)
            evaluate_plan(plan, tt, xs[:256])  # warm-up
            reps, t_cpu = 0, 0.0
            while reps < 5 and t_cpu < 15.0:
                t1 = time.perf_counter()
                evaluate_plan(plan, tt, xs)
                t_cpu += time.perf_counter() - t1
                reps += 1
            result["cpu_baseline"] = {
                "value": reps * B / t_cpu,
                "unit": "evals/s",
                "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"{reps} x one {B}-row batch of the same workload through oracle/torch_oracle.py "
                          "(op-for-op restatement of the reference's torch-CPU forward, fp32, no_grad)",
            }
        result["variants"] = variants
        if world == 1 and not args.no_other_configs:
            result["other_configs"] = other_configs(device, stream, B)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
