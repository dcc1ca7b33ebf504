#!/usr/bin/env python3
"""Headline benchmark: log-likelihood evaluations / second of the folded log-space forward.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]; configs[2] at N=8): 784-variable QuadTree-2 circuit,
Categorical-256 inputs, CP sum layers, K=32, fp32, batch 4096 PER GPU (weak scaling), synthetic
int64 batches resident in HBM, closed-form random-init parameters (cirkit_amd.initializers).
One "step" = one forward of the rank's 4096-row batch (parameter softmax/log recomputed every step,
as the reference does) + the device-side sum of the log-likelihoods (+ for N > 1 the exchange of its [sum, count]
pair: RCCL all-reduces carrying the pairs of up to 64 consecutive steps each, all completed inside the timed region).

K timed steps rotate through 12 distinct resident batches (308 MB of int64, more than the 256 MB Infinity Cache); the
timed region is repeated `--rounds` times (each round: exactly K steps between barrier + synchronize) and the MEDIAN
round is reported; by default there are at least 5 rounds and enough of them for ~400 steps in all, because the device
keeps speeding up over its first ~50 ms of work; `--settle` untimed steps precede every timed region -- `timing` lists every round.

Prints ONE JSON line on rank 0 (contract in the task description), with
  roofline     -- the dominant kernel (by HIP-event time per launch, measured here): the contraction flops it EXECUTES
                  against the dense fp32-input MFMA peak (157.3 TFLOP/s), or its measured HBM bytes against 8 TB/s,
                  whichever binds; `traffic` / `mfma_busy_frac` from three short rocprofv3 counter passes run by this
                  very invocation (FETCH_SIZE x 2 + WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES; `traffic_source` says so, or names
                  the committed profile when rocprofv3 cannot be run); `roofline.forward` = the whole step by
                  algorithmic bytes (SURVEY.md section 8 d); `roofline.algorithmic` = the same convention per launch;
  check        -- mean LL of the last step and the max relative error of 64 rows against the CPU oracle;
  cpu_baseline -- the CPU oracle (op-for-op port of the reference's torch-CPU path) timed on the
                  host cores of this box on a bounded sample (rank 0, N = 1 only); host core count and CPU model stated.
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


def contraction_flops(plan, B: int, real_products: int = 1) -> float:
    """The fp32 contraction flops one forward of `plan` executes on B rows: 2 Ko Ki per row and fold of every dense / CP-T fold
    (x H for a dense layer over concatenated children); a mixing layer is an element-wise weighted sum (no contraction), a
    collapsed Sum -> Sum pair (MatMul weight) is evaluated as the mixing + dense pair it was.  `real_products`: real contractions
    per complex one (1 for real circuits and for real-parameter circuits on signed tiles, 4 on the complex kernels)."""
    total = 0.0
    for l in plan.layers:
        if l.type not in ("sum", "cpt", "tucker", "tensordot"):
            continue
        ops = [op for pg in l.params.values() for op in pg.ops]
        per_fold = 2.0 * l.num_output_units * l.num_input_units
        if l.type == "tucker":
            per_fold *= l.num_input_units ** (l.arity - 1)
        elif l.type == "sum" and l.arity > 1:
            if "mixing_weight" in ops and "matmul" not in ops:
                continue
            if "mixing_weight" not in ops:
                per_fold *= l.arity
        total += per_fold * l.num_folds * B
    return total * real_products


def bench_summary(result: dict) -> dict:
    """The numbers DESIGN.md section 8 quotes, compact, as the LAST key of the line (the driver's record keeps the line's tail)."""
    oc = result.get("other_configs") or {}

    def pick(d, ms_key):
        if not d or ms_key not in d:
            return None
        o = {"ms": round(float(d[ms_key]), 4)}
        if d.get("frac_of_fp32_mfma") is not None:
            o["mfma"] = round(float(d["frac_of_fp32_mfma"]), 3)
        return o

    roof = result.get("roofline") or {}
    s = {
        "cfg2_fwd": {"ms": round(float(result["ms_per_step"]), 4), "mfma": (round(float(roof["frac"]), 3) if roof.get("frac") is not None else None)},
        "cfg4_fwd": pick(oc.get("config4"), "ms_per_forward"),
        "cfg5_fwd": pick(oc.get("config5"), "ms_per_forward"),
        "cfg5_Z": ({"ms": round(float(oc["config5"]["ms_partition_function"]), 4)} if "config5" in oc else None),
        "cfg5_complex_fwd": pick(oc.get("config5_complex_weights"), "ms_per_forward"),
        "tucker_nb_fwd": pick(oc.get("notebook_quadgraph_tucker_k64_b128"), "ms_per_forward"),
        "train_cfg2": pick(oc.get("train_step_cfg2"), "ms_per_step"),
        "train_nb_k64_b256": pick(oc.get("train_step_notebook_quadgraph_cp_k64_b256"), "ms_per_step"),
        "train_cfg4_b1024": pick(oc.get("train_step_cfg4_b1024"), "ms_per_step"),
        "train_cfg5_sq": pick(oc.get("train_step_cfg5_squared"), "ms_per_step"),
        "cpu_evals_s": (round(float(result["cpu_baseline"]["value"]), 1) if "cpu_baseline" in result else None),
        "dist": (result.get("distributed") or {}).get("backend"),
        "note": "ms per forward / per training step; mfma = executed contraction flops / time / 157.3 TFLOP/s (fp32-input MFMA peak)",
    }
    return {k: v for k, v in s.items() if v is not None}


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
    t4 = init_plan_tensors(plan4)
    x4 = torch.randn((B, 784), generator=g).to(device)
    hc = HipCircuit(plan4, t4, device=device)
    ms = time_forward(hc, x4)
    y4 = hc(x4).double().cpu()
    alg = plan4.algorithmic_bytes(B)["total"]
    out["config4"] = {
        "workload": f"Poon-Domingos 28x28 (delta 4), Gaussian leaves, CP sum layers, mixing layers, K=64, batch {B}, "
                    "38 folded layers; CP blocks + mixing layers fused per region (cirkit_amd/csrc/ck_cp.hip)",
        "ms_per_forward": ms, "evals_per_s": B / ms * 1e3, "algorithmic_bytes": alg,
        "hbm_roofline_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "executed_flops": contraction_flops(plan4, B),
        "frac_of_fp32_mfma": contraction_flops(plan4, B) / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
        "variants": {},
    }
    del hc
    # Labelled variants, never `ms_per_forward`: the region / CP-block launches on the bf16 matrix instructions (a weight unit cut
    # into 2 / 3 bf16 pieces in LDS, the exponentiated tile in registers; fp32 accumulation: ck_region_lse_fwd_v, ck_cp_lse_fwd_v).
    for cname in ("bf16x3", "bf16x6"):
        hv = HipCircuit(plan4, t4, device=device, contraction=cname)
        ms_v = time_forward(hv, x4)
        y_v = hv(x4).double().cpu()
        out["config4"]["variants"][f"contraction={cname}"] = {
            "ms_per_forward": ms_v, "evals_per_s": B / ms_v * 1e3,
            "max_rel_diff_to_exact_fp32": float(((y_v - y4).abs() / y4.abs()).max()),
            "what": ("K = 64 contractions of the region / CP-block launches on v_mfma_f32_32x32x16_bf16: operands split by truncation into "
                     + ("2 bf16 pieces, 3 products (~2^-15 per product)" if cname == "bf16x3" else "3 bf16 pieces, 6 products (fp32-like)")
                     + "; everything else exact fp32"),
        }
        del hv
    plan5 = image_data((1, 28, 28), "quad-tree-2", input_layer="embedding", num_input_units=32, sum_product_layer="cp-t",
                       num_sum_units=32, sum_weight_activation="none", semiring="complex-lse-sum")
    t5 = init_plan_tensors(plan5)
    hc = HipCircuit(plan5, t5, device=device)
    ms = time_forward(hc, torch.randint(0, 256, (B, 784), generator=g).to(device))
    alg = plan5.algorithmic_bytes(B)["total"]
    hz = HipCircuit(squared_partition_plan(plan5), hc.store, device=device)
    ms_z = time_forward(hz, None)
    out["config5"] = {
        "workload": f"squared (SoS) circuit: QuadTree-2 28x28, Embedding-256, CP-T, K=32, complex-lse-sum, batch {B}, REAL parameters "
                    "(BASELINE config 5 as SURVEY 8(d) words it): evaluated on signed linear tiles -- the reference's complex "
                    "logarithms of real numbers; complex-valued weights take the complex kernels, ~0.6 ms); "
                    "Z = integral |c|^2 built from the plan of c (cirkit_amd/functional.py)",
        "ms_per_forward": ms, "evals_per_s": B / ms * 1e3, "algorithmic_bytes": alg,
        "hbm_roofline_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "executed_flops": contraction_flops(plan5, B),  # signed tiles: one real contraction per complex one
        "frac_of_fp32_mfma": contraction_flops(plan5, B) / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
        "ms_partition_function": ms_z,
    }
    del hc, hz
    # ... and on the complex path (what a circuit with complex-valued parameters takes; `signed_real=False` forces it for
    # these real parameters): linear (re, im) tiles, two fp32 MFMA chains per contraction
    hq = HipCircuit(plan5, t5, device=device, signed_real=False)
    ms_q = time_forward(hq, torch.randint(0, 256, (B, 784), generator=g).to(device))
    out["config5_complex_weights"] = {
        "workload": "the same circuit evaluated as complex-valued parameters require (signed_real=False): values as (re + i im) 2^e tiles, "
                    "Embedding -> 3 CP-T levels in one launch, the complex logarithm once at the output (profiles/r06_b_cfg5_complex.txt; "
                    "0.575 ms on the layer-wise (log|v|, arg v) kernels of round 5)",
        "ms_per_forward": ms_q, "evals_per_s": B / ms_q * 1e3, "launches": hq.num_launches(B),
        # real weights on complex values: two real (32, 32) contractions per layer and tile (W re, W im); four with complex weights
        "executed_flops": contraction_flops(plan5, B, 2),
        "frac_of_fp32_mfma": contraction_flops(plan5, B, 2) / (ms_q * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
        "path": ("linear (re, im) tiles (cirkit_amd/circuit_clin.py, csrc/ck_clin.hip)" if getattr(hq, "_clin", None) is not None
                 else "layer-wise (log|v|, arg v) kernels"),
    }
    del hq
    out["train_step_cfg2"] = train_step_cfg2(device, stream, B)
    # the reference's own training loop (notebooks/learning-a-circuit.ipynb cells 4 / 16 / 18: QuadGraph, CP, K = 64, batch 256,
    # Adam) and BASELINE config 4 at 1024 rows: the job form of the training step (cirkit_amd/train_jobs.py)
    out["train_step_notebook_quadgraph_cp_k64_b256"] = train_step_k64(device, stream, "notebook", 256)
    out["train_step_cfg4_b1024"] = train_step_k64(device, stream, "cfg4", 1024)
    # BASELINE config 5 trained: loss = -mean(2 Re c(x) - Re Z), Adam, every parameter graph of c and of Z evaluated every step
    out["train_step_cfg5_squared"] = train_step_squared(device, stream, plan5, t5, B)
    # The only forward timing the reference publishes (notebooks/compilation-options.ipynb:594):
    # QuadGraph 28x28, Categorical-256, Tucker layers, K = 64, batch 128, fold + optimize: 38.6 ms on an unnamed
    # CUDA GPU.  Different hardware and not the north-star metric, so it stays out of `vs_baseline`.
    plan_nb = image_data((1, 28, 28), "quad-graph", input_layer="categorical", num_input_units=64,
                         sum_product_layer="tucker", num_sum_units=64)
    t_nb = init_plan_tensors(plan_nb)
    x_nb = torch.randint(0, 256, (128, 784), generator=g).to(device)
    hc = HipCircuit(plan_nb, t_nb, device=device)
    ms = time_forward(hc, x_nb)
    y_exact = hc(x_nb).double().cpu()
    out["notebook_quadgraph_tucker_k64_b128"] = {
        "workload": "QuadGraph 28x28, Categorical-256, Tucker, K=64, batch 128 (the reference's compilation-options "
                    "notebook); Tucker layers on MFMA (cirkit_amd/csrc/ck_gemm.hip)",
        "ms_per_forward": ms, "evals_per_s": 128 / ms * 1e3,
        "reference_published_ms": 38.6, "reference_hardware": "unnamed CUDA GPU (notebook output)",
        "executed_flops": contraction_flops(plan_nb, 128),
        "frac_of_fp32_mfma": contraction_flops(plan_nb, 128) / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
        "variants": {},
    }
    del hc
    # Labelled variants, never `ms_per_forward`: the stream-K Tucker launch on the bf16 matrix instructions (the staged weights
    # and e_r cut into 2 / 3 bf16 pieces, 3 / 6 products per 16 right indices, fp32 accumulation: ck_tucker_fwd).
    for cname in ("bf16x3", "bf16x6"):
        hv = HipCircuit(plan_nb, t_nb, device=device, contraction=cname)
        ms_v = time_forward(hv, x_nb)
        y_v = hv(x_nb).double().cpu()
        out["notebook_quadgraph_tucker_k64_b128"]["variants"][f"contraction={cname}"] = {
            "ms_per_forward": ms_v, "evals_per_s": 128 / ms_v * 1e3,
            "max_rel_diff_to_exact_fp32": float(((y_v - y_exact).abs() / y_exact.abs()).max()),
            "what": ("Tucker contractions on v_mfma_f32_32x32x16_bf16: operands split by truncation into "
                     + ("2 bf16 pieces, 3 products (~2^-15 per product)" if cname == "bf16x3" else "3 bf16 pieces, 6 products (fp32-like)")
                     + "; everything else exact fp32"),
        }
        del hv
    return out


def train_step_cfg2(device, stream, B: int, rounds: int = 5, steps: int = 40, settle_s: float = 0.3) -> dict:
    """One maximum-likelihood training step at BASELINE config 2 -- forward, backward, Adam, parameters re-evaluated every step
    (the reference's loop: notebooks/learning-a-circuit.ipynb cells 18 / 20) -- through `HipTrainer`'s fused form: measured like
    the headline forward (clocks settled first, `rounds` rounds of `steps` steps between HIP events on the launch stream, the
    median round), int64 batches rotating over more than the 256 MB Infinity Cache.  executed_flops: the fp32 MFMA
    contractions the step issues (forward levels + 2 per node backward, tail forward + 3 per fold backward, the dense layer
    on the (C + 1)-row table forward + 3 backward)."""
    import time

    import numpy as np
    import torch

    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data
    from cirkit_amd.training import HipTrainer

    plan = image_data((1, 28, 28), "quad-tree-2", input_layer="categorical", num_input_units=32,
                      sum_product_layer="cp", num_sum_units=32)
    g = torch.Generator().manual_seed(11)
    xs = [torch.randint(0, 256, (B, 784), generator=g).to(device) for _ in range(12)]
    with torch.cuda.stream(stream):
        tr = HipTrainer(plan, init_plan_tensors(plan), device=device, lr=0.01, optimizer="adam")
        k = 0
        t0 = time.perf_counter()
        first = None
        while time.perf_counter() - t0 < settle_s or k < 10:
            ll = tr.step(xs[k % 12])
            if first is None:
                first = ll.clone()
            k += 1
        torch.cuda.synchronize(device)
        per_round = []
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(steps):
                ll = tr.step(xs[k % 12])
                k += 1
            b.record(stream)
            torch.cuda.synchronize(device)
            per_round.append(a.elapsed_time(b) / steps)
        last = ll.clone()
    ms = float(np.median(per_round))
    fused = bool(tr.fused)
    c = tr.circuit
    chains = 0.0  # 32 x 32 x 32-row contractions
    if fused:
        grp = tr._fz["group"]
        tiles = (B + 31) // 32
        chains += sum(c.layers[j].num_folds for j in grp.levels) * tiles * 3
        chains += sum(c.layers[j].num_folds for j in c._tail if c.layers[j].num_output_units == 32) * tiles * 4
        chains += c.layers[grp.dense_layer].num_folds * ((c.layers[grp.input_layer].num_categories + 1 + 31) // 32) * 4
    flops = chains * 2.0 * 32 * 32 * 32
    return {
        "workload": f"784-var QuadTree (QT-2) PC, Categorical-256 leaves, K=32, batch {B}: forward + backward + Adam, parameters "
                    "re-evaluated every step; fused training step (cirkit_amd/training.py, ck_leaf_walk_fwd keep_levels + ck_leaf_walk_bwd), the optimizer "
                    "and the next step's parameters in the backward epilogues (ck_table_dense_bwd / ck_param_softmax_bwd_batch with opt)",
        "fused": fused, "ms_per_step": ms, "samples_per_s": B / ms * 1e3, "ms_per_step_by_round": per_round,
        "settle_steps": k - rounds * steps, "steps_timed_total": rounds * steps,
        "executed_flops": flops, "frac_of_fp32_mfma": flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
        "mean_ll_first_step": float(first[0] / first[1]), "mean_ll_last_step": float(last[0] / last[1]),
        "optimizer": "adam(lr=0.01), in the backward epilogues, device-side clock" if tr._fused_opt_ok() else "adam(lr=0.01)", "dtype": "f32",
    }


def train_step_k64(device, stream, which: str, B: int, rounds: int = 5, steps: int = 30, settle_s: float = 0.3) -> dict:
    """One maximum-likelihood training step of a 64-unit CP circuit -- forward, backward, Adam, every parameter graph evaluated
    every step -- through `HipTrainer`'s job form (level launches over jobs, the optimizer in the job epilogues): `which` =
    "notebook": the circuit the reference trains in notebooks/learning-a-circuit.ipynb (QuadGraph 28x28, Categorical-256, CP,
    K = 64; cell 16: batch 256, cell 18: Adam(lr=0.01)); "cfg4": BASELINE config 4 (Poon-Domingos 28x28, Gaussian leaves, CP,
    K = 64).  Measured like `train_step_cfg2`.  executed_flops: the fp32 MFMA contractions of the sum jobs (1 per 64 x 64 x 32-row
    tile forward, 3 backward)."""
    import time

    import numpy as np
    import torch

    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data
    from cirkit_amd.training import HipTrainer

    g = torch.Generator().manual_seed(13)
    if which == "cfg4":
        plan = image_data((1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64, sum_product_layer="cp", num_sum_units=64)
        xs = [torch.randn((B, 784), generator=g).to(device) for _ in range(12)]
        workload = f"Poon-Domingos 28x28, Gaussian leaves, CP sum layers, K=64 (BASELINE config 4), batch {B}"
    else:
        plan = image_data((1, 28, 28), "quad-graph", input_layer="categorical", num_input_units=64, sum_product_layer="cp", num_sum_units=64)
        xs = [torch.randint(0, 256, (B, 784), generator=g).to(device) for _ in range(12)]
        workload = f"QuadGraph 28x28, Categorical-256, CP, K=64, batch {B} (the reference's learning-a-circuit notebook)"
    with torch.cuda.stream(stream):
        tr = HipTrainer(plan, init_plan_tensors(plan), device=device, lr=0.01, optimizer="adam")
        k, t0, first = 0, time.perf_counter(), None
        while time.perf_counter() - t0 < settle_s or k < 10:
            ll = tr.step(xs[k % 12])
            if first is None:
                first = ll.clone()
            k += 1
        torch.cuda.synchronize(device)
        per_round = []
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(steps):
                ll = tr.step(xs[k % 12])
                k += 1
            b.record(stream)
            torch.cuda.synchronize(device)
            per_round.append(a.elapsed_time(b) / steps)
        last = ll.clone()
    ms = float(np.median(per_round))
    js = tr._jobs
    tiles = (B + 31) // 32
    flops = (len(js.sum_jobs) * tiles * 4 * 2.0 * 64 * 64 * 32) if js is not None else 0.0
    return {
        "workload": workload + ": forward + backward + Adam, parameters re-evaluated every step",
        "form": "job list (cirkit_amd/train_jobs.py, csrc/ck_jobs.hip)" if js is not None else "layer-wise launch list",
        "launches_per_step": js.num_launches(B) if js is not None else None,
        "sum_jobs": len(js.sum_jobs) if js is not None else None, "mix_jobs": len(js.mix_jobs) if js is not None else None,
        "ms_per_step": ms, "samples_per_s": B / ms * 1e3, "ms_per_step_by_round": per_round,
        "settle_steps": k - rounds * steps, "steps_timed_total": rounds * steps,
        "executed_flops": flops, "frac_of_fp32_mfma": flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
        "mean_ll_first_step": float(first[0] / first[1]), "mean_ll_last_step": float(last[0] / last[1]),
        "optimizer": "adam(lr=0.01), in the job epilogues", "dtype": "f32",
    }


def train_step_squared(device, stream, plan_c, tensors, B: int, rounds: int = 5, steps: int = 20, settle_s: float = 0.3) -> dict:
    """One maximum-likelihood step of the squared circuit of BASELINE config 5 through `HipSquaredTrainer`: c(x) on signed-log
    blocks (csrc/ck_signed.hip), Z = integral |c|^2 (ConstantValue / Hadamard / TensorDot layers over Gram matrices of the
    Embedding weights) forward and backward beside it on a second stream, one optimizer launch with a device clock; three
    recorded launch lists replayed by the native executor.  Measured like `train_step_cfg2`."""
    import time

    import numpy as np
    import torch

    from cirkit_amd.training_squared import HipSquaredTrainer

    g = torch.Generator().manual_seed(17)
    # (the counter-based initializer emits a few exact zeros: log 0 under autograd is NaN in the reference as well)
    tensors = {k: np.where(np.asarray(v) == 0, np.float32(1e-2), np.asarray(v)).astype(np.float32) for k, v in tensors.items()}
    xs = [torch.randint(0, 256, (B, 784), generator=g).to(device) for _ in range(12)]
    with torch.cuda.stream(stream):
        tr = HipSquaredTrainer(plan_c, tensors, device=device, lr=1e-3, optimizer="adam")
        k, t0, first = 0, time.perf_counter(), None
        while time.perf_counter() - t0 < settle_s or k < 10:
            ll = tr.step(xs[k % 12])
            if first is None:
                first = ll.clone()
            k += 1
        torch.cuda.synchronize(device)
        per_round = []
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(steps):
                ll = tr.step(xs[k % 12])
                k += 1
            b.record(stream)
            torch.cuda.synchronize(device)
            per_round.append(a.elapsed_time(b) / steps)
        last = ll.clone()
        taken, dropped = tr.opt_counters()
    ms = float(np.median(per_round))
    return {
        "workload": f"BASELINE config 5 (squared QuadTree-2 28x28, Embedding-256, CP-T, K=32, real parameters), batch {B}: forward + backward "
                    "of c and of Z + Adam, parameters re-evaluated every step",
        "form": ("c on signed-log blocks (fp32 log|v| + a sign bit, ck_signed.hip)" if tr._signed is not None else "c on complex layer-wise kernels")
                + "; Z (Hadamard layers as lists, TensorDot pairs in one launch) on a second stream; three recorded launch lists",
        "ms_per_step": ms, "samples_per_s": B / ms * 1e3, "ms_per_step_by_round": per_round,
        "settle_steps": k - rounds * steps, "steps_timed_total": rounds * steps, "optimizer_steps_taken": taken, "steps_dropped": dropped,
        "mean_ll_first_step": float(first[0] / first[1]), "mean_ll_last_step": float(last[0] / last[1]),
        # c's contractions: 1 forward + 3 backward (y again, W^T t, dW) per fold and tile; Z's one-row layers are not counted
        "executed_flops": 4.0 * contraction_flops(plan_c, B),
        "frac_of_fp32_mfma": 4.0 * contraction_flops(plan_c, B) / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
        "optimizer": "adam(lr=0.001), one launch on the flat parameter buffer, device-side clock", "dtype": "f32",
    }


def _free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo", encoding="utf-8") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def live_pmc(argv_child: list[str], timeout_s: float = 200.0) -> dict | None:
    """HBM bytes and MFMA-busy cycles per launch of every kernel of a step, measured NOW: three short rocprofv3 passes
    (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_VALU_MFMA_BUSY_CYCLES`; one counter per pass, kernel trace only --
    MI355X_MICROARCH.md, HBM / rocprofv3 sections) over `bench.py --pmc-child` (the same circuit and inputs, 12 steps).
    FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes); both sizes are reported in KiB by rocprofv3.
    Returns {kernel: {"read_bytes", "write_bytes", "hbm_bytes", "mfma_busy_cycles"}} or None when rocprofv3 is not
    usable here (missing, nested under another profiler, timeout)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or os.environ.get("CIRKIT_BENCH_NO_PMC"):
        return None
    out: dict[str, dict] = {}
    t_end = time.time() + timeout_s

    def short_name(name: str) -> str:
        return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]

    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp", CIRKIT_BENCH_NO_PMC="1")
        # rank 0 of an N > 1 run profiles a plain single-process child on its own device (the other ranks wait at the
        # closing barrier): nothing of the launcher's rendezvous may reach it
        for k in list(env):
            if k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "MASTER_ADDR",
                     "MASTER_PORT", "BENCH_FORCE_DIST") or k.startswith(("TORCHELASTIC_", "GROUP_WORLD_", "ROLE_WORLD_")):
                env.pop(k)
        # pass 0: kernel trace alone (no counters): the launch durations the profiler sees, warm -- 1500 steps, the last
        # 50 dispatches of every kernel averaged.  HIP events around every launch (the instrumented pass of
        # profile_kernels) stretch a launch by a few microseconds; this is the figure a rocprofv3 --stats summary gives.
        d = os.path.join(tmp, "trace")
        cmd = [exe, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--pmc-child", "--pmc-steps", "1500", *argv_child]  # (the clocks take ~0.1 s to settle: the first rounds of the
        # timed region are ~10 % slower too -- `first_round_ms_per_step`)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=max(10.0, t_end - time.time()), check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("results.db")]
            con = sqlite3.connect(dbs[0])
            per: dict[str, list[float]] = {}
            for name, start, end in con.execute("select name, start, end from kernels order by start"):
                per.setdefault(short_name(name), []).append(float(end - start))
            con.close()
            for short, durs in per.items():
                if short.startswith(("__amd", "at::")) or len(durs) < 60:
                    continue
                tail_d = durs[-50:]
                out.setdefault(short, {})["trace_us"] = sum(tail_d) / len(tail_d) / 1e3
        except Exception:  # noqa: BLE001 -- the trace figure is optional
            pass
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child", *argv_child]
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=max(10.0, t_end - time.time()), check=True)
                dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("results.db")]
                con = sqlite3.connect(dbs[0])
                rows = list(con.execute("select kernel_name, count(*), avg(value) from counters_collection "
                                        "where counter_name = ? group by kernel_name", (ctr,)))
                con.close()
            except Exception:  # noqa: BLE001 -- any failure: no live numbers
                return None
            for name, n, avg in rows:
                short = short_name(name)
                if short.startswith(("__amd", "at::")):
                    continue
                e = out.setdefault(short, {})
                e["launches_sampled"] = int(n)
                if ctr == "FETCH_SIZE":
                    # x 2: measured for every access kind of these kernels -- coalesced streams, 128- and 256-byte row gathers, through
                    # registers and through LDS DMA alike (scripts/ubench/fetch_calib.hip, profiles/r05_fetch_calib.txt: 2.000 each)
                    e["read_bytes"] = 2.0 * avg * 1024.0
                    e["read_bytes_raw"] = avg * 1024.0
                elif ctr == "WRITE_SIZE":
                    e["write_bytes"] = avg * 1024.0
                else:
                    e["mfma_busy_cycles"] = avg
    for e in out.values():
        if "read_bytes" in e and "write_bytes" in e:
            e["hbm_bytes"] = e["read_bytes"] + e["write_bytes"]
            e["hbm_bytes_raw"] = e["read_bytes_raw"] + e["write_bytes"]
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=0,
                    help="rounds of --steps timed steps; the median round is reported.  Default: at least 5 and enough for ~400 "
                         "steps in all -- the device keeps speeding up over the first ~150 steps (0.135 -> 0.123 ms per step, "
                         "measured round by round), so with the driver's 20 steps per round the median of 5 rounds would still "
                         "be a warm-up round")
    ap.add_argument("--batches", type=int, default=12,
                    help="distinct resident input batches the steps rotate through (12 x 25.7 MB of int64 > the 256 MB "
                         "Infinity Cache, so no step finds its input cached)")
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="rows per GPU (default: the BASELINE config)")
    ap.add_argument("--no-graph", action="store_true", help="never replay as a hipGraph (only lists of more than 64 launches are by default)")
    ap.add_argument("--fuse", type=int, default=-1, help="-1: full leaf fusion (default), 0: layer-wise, n: n CP-T levels")
    ap.add_argument("--contraction", default="f32", choices=["f32"], help="K=32 sum layers: exact fp32 MFMA (the only form)")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary measurements (cached parameters, parameters at the start, two streams)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short measurements of BASELINE configs 4 and 5 (never part of `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-breakdown", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the three short rocprofv3 counter passes (roofline.traffic then comes from profiles/)")
    ap.add_argument("--dist", action="store_true",
                    help="take the distributed code path (process group + the bucketed all-reduce of every step's [sum, count]) even at world size 1")
    ap.add_argument("--params-at-start", action="store_true",
                    help="evaluate the parameter graphs with a launch of their own at the START of every forward "
                         "(HipCircuit(params_at_end=False)) instead of inside the launch that walks the tail of the forward "
                         "before (the default; every step evaluates them once either way)")
    ap.add_argument("--params-at-end", action="store_true", help=argparse.SUPPRESS)  # (the default now; kept for old command lines)
    ap.add_argument("--settle", type=int, default=2000,
                    help="untimed steps in front of every timed region beside --warmup (the clocks settle over ~50 ms of work)")
    ap.add_argument("--staged-input", action="store_true",
                    help="stage the batch (int64 (B, D) -> int32 (D, B)) with a launch of its own, as in round 2, instead of "
                         "letting the leaf launch read the caller's tensor")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="exchange the [sum, count] pairs through torch.distributed (bucketed, 64 steps per collective) instead of "
                         "the library's own RCCL communicator (ck_comm_*, one collective per step on the launch stream)")
    ap.add_argument("--sync-collectives", action="store_true",
                    help="with the library's communicator: the all-reduce of every step ON the launch stream (the next step starts "
                         "behind it) instead of on the communicator's own stream beside the next step")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-steps", type=int, default=12, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as invoked by hand / by the driver: become the launcher -- one rank per GPU under
        # torch.distributed.run on this node (rendezvous on 127.0.0.1), same arguments; rank 0 prints the JSON line
        import subprocess

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

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
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    # one process per GPU; BENCH_DIST_BACKEND=gloo lets the N > 1 path be exercised on a 1-GPU box
    # (ranks then share the device; RCCL itself refuses two ranks on one GPU).  `--dist` (or BENCH_FORCE_DIST=1) takes
    # the distributed code path -- process group, ring of async all-reduces, barriers, MAX over ranks -- at world size 1
    # too: that is how RCCL itself is exercised on a 1-GPU box (tests/test_gpu_bench_distributed.py).
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    use_dist = world > 1 or args.dist or os.environ.get("BENCH_FORCE_DIST") == "1"
    device = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(device)
    ranks_seen = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # every rank contributes 1 through the collective the bench uses: the sum is the number of ranks the backend saw
        one = torch.ones(1, dtype=torch.float64, device=device)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())
    # The exchange of the data path goes through the C ABI: an RCCL communicator owned by libcirkit_hip (ck_comm_*), its
    # all-reduce enqueued on the launch stream right behind the launch that writes the pair -- one collective per step, no
    # host work in between.  torch.distributed only carried the 128-byte ncclUniqueId (and keeps the barriers / the MAX of
    # the wall clocks, which are not on the data path).  BENCH_DIST_BACKEND=gloo (several ranks on ONE device, which RCCL
    # refuses) keeps torch.distributed for the exchange, bucketed as before.
    comm, comm_error = None, None
    if use_dist and backend == "nccl" and not args.torch_collectives:
        from cirkit_amd.distributed import HipComm, set_default_comm

        try:
            comm = HipComm.from_process_group(device)
            set_default_comm(comm)
            probe = torch.ones(1, dtype=torch.float64, device=device)
            comm.all_reduce(probe)
            ranks_seen = int(probe.item())
        except Exception as e:  # noqa: BLE001 -- e.g. no librccl to bind: the exchange stays on torch.distributed (reported)
            comm, comm_error = None, f"{type(e).__name__}: {e}"
            set_default_comm(None)
        # (every rank must take the same path: a rank whose communicator failed makes all of them fall back)
        ok = torch.tensor([1.0 if comm is not None else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if comm is not None and float(ok.item()) == 0.0:
            comm.destroy()
            comm, comm_error = None, "another rank could not create its communicator"
            set_default_comm(None)

    # BASELINE configs[1], built natively (cirkit_amd/templates.py; identical to the plan the reference
    # compiles -- tests/test_templates.py pins it against the committed reference fixture)
    plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32,
                      sum_product_layer="cp", num_sum_units=32)
    tensors = init_plan_tensors(plan)
    B = args.batch
    fuse = True if args.fuse < 0 else (False if args.fuse == 0 else args.fuse)
    circuit = HipCircuit(plan, tensors, device=device, use_graph=not args.no_graph, fuse=fuse,
                         contraction=args.contraction, direct_input=not args.staged_input,
                         params_at_end=not args.params_at_start)
    g = torch.Generator().manual_seed(1234 + rank)
    nb = max(1, args.batches)
    xs = [torch.randint(0, 256, (B, plan.num_variables), generator=g).to(device) for _ in range(nb)]  # int64, like the reference
    x = xs[0]

    stream = torch.cuda.Stream(device)

    if args.pmc_child:  # profiled by live_pmc(): the same steps, nothing else
        with torch.cuda.stream(stream):
            for k in range(max(1, args.pmc_steps)):
                circuit.log_likelihood_sum(xs[k % nb])
        torch.cuda.synchronize(device)
        return

    # The device's clocks keep rising over its first ~50 ms of work (0.113 -> 0.101 ms per step from the first round of 50
    # steps to the seventh, `profiles/r03_d_bench.json`): beside the W warm-up steps every timed region is preceded by
    # `--settle` untimed steps (0.2 s by default), reported as `timing.settle_steps`, so that the rounds measure one state.
    settle_steps = max(0, int(args.settle))
    rank_spread: list = []  # N > 1: (fastest, slowest) rank wall time of every timed round of the last timed region

    def timed_region(circ, steps, warmup, rounds=1, settle=None):
        """W untimed steps, then `rounds` rounds of exactly K timed steps, each round with a barrier + synchronize on both
        sides.  Returns (wall seconds per round -- max over ranks --, HIP-event ms per step per round on the launch
        stream, the [sum, count] over ranks of the last step)."""
        last = [None]
        # N > 1: every step's [sum, count] pair is exchanged, in BUCKETS: step k copies its pair into row k % BUCKET of a
        # (BUCKET, 2) buffer and one all-reduce per full bucket (and one at the end of a round) carries them all -- fewer,
        # larger collectives (a collective per step cost 15 us of every 102 us step at world size 1: its own launch, an event
        # pair between the streams, and the host side of the call).  Two buffers alternate: the collective of one runs on
        # RCCL's stream while the steps fill the other; a buffer is reused only after its previous collective has completed.
        BUCKET = 64
        bufs = [torch.zeros((BUCKET, 2), dtype=torch.float64, device=device) for _ in range(2)] if use_dist else []
        works: list = [None, None]
        cur = [0, 0]  # (buffer in use, rows filled)
        fed = [0]

        def flush() -> None:
            i, n = cur
            if n:
                works[i] = dist.all_reduce(bufs[i][:n], op=dist.ReduceOp.SUM, async_op=True)  # the exchange: n x 16 bytes over xGMI
                cur[0], cur[1] = 1 - i, 0
                if works[1 - i] is not None:
                    works[1 - i].wait()  # (stream-level: the buffer about to be refilled has been reduced)
                    works[1 - i] = None

        RING = 64
        ring = torch.zeros((RING, 2), dtype=torch.float64, device=device) if comm is not None else None

        def step() -> None:
            x = xs[fed[0] % nb]
            fed[0] += 1
            if comm is not None:
                # forward + device-side sum on `stream`; the all-reduce of the pair on the communicator's own stream, ordered
                # behind this step (ck_comm_all_reduce_async_f64): nothing of the next step waits for it.  A row of the ring is
                # written again RING steps later: every RING / 2 steps the launch stream waits (on the device) for the
                # collectives issued so far
                row = ring[fed[0] % RING]
                last[0] = circ.log_likelihood_sum(x, out=row)
                if args.sync_collectives:
                    comm.all_reduce(row)
                else:
                    comm.all_reduce_async(row)
                    if fed[0] % (RING // 2) == 0:
                        comm.wait()
            elif use_dist:
                i, n = cur
                # forward + device-side sum, enqueued on `stream`; the launch that ends the forward writes the pair into its row
                last[0] = circ.log_likelihood_sum(x, out=bufs[i][n])
                cur[1] = n + 1
                if cur[1] == BUCKET:
                    flush()
            else:
                last[0] = circ.log_likelihood_sum(x)

        def drain() -> None:
            if comm is not None and not args.sync_collectives:
                comm.wait()  # (before the closing event: every collective of the timed steps has run when the clock stops)
            if use_dist and comm is None:
                flush()
                for i in range(2):
                    if works[i] is not None:
                        works[i].wait()
                        works[i] = None

        walls, evms = [], []
        rank_spread.clear()
        with torch.cuda.stream(stream):
            for _ in range(warmup + (settle_steps if settle is None else settle)):
                step()
            drain()
            for _ in range(rounds):
                torch.cuda.synchronize(device)
                if use_dist:
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
                if use_dist:
                    dist.barrier()
                torch.cuda.synchronize(device)
                wall = time.perf_counter() - t0
                if use_dist:
                    t = torch.tensor([wall, -wall], dtype=torch.float64, device=device)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    rank_spread.append((-float(t[1].item()), float(t[0].item())))  # (fastest, slowest rank of this round)
                    wall = float(t[0].item())
                walls.append(wall)
                evms.append(e0.elapsed_time(e1) / steps)
        pair = last[0].cpu()
        return walls, evms, pair

    if args.rounds <= 0:
        args.rounds = max(5, -(-400 // max(1, args.steps)))
    # the literal protocol first -- W warm-up steps, then K timed steps, nothing else before them: cold clocks
    cold_walls, _, _ = timed_region(circuit, args.steps, args.warmup, 1, settle=0)
    walls, evms, pair = timed_region(circuit, args.steps, args.warmup, max(1, args.rounds))
    spread = list(rank_spread)
    elapsed = float(np.median(walls))  # the median round (each round: exactly K steps between barriers)
    step_ms_events = float(np.median(evms))
    total_nll = float(pair[0])
    total_rows = float(pair[1])

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
        "steps_timed_total": len(walls) * args.steps,
        "first_round_ms_per_step": 1e3 * walls[0] / args.steps,
        "cold_first_round_ms_per_step": 1e3 * cold_walls[0] / args.steps,  # `--warmup W --steps K` literally: no settle steps in front
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
            # derived from what a step launches: the parameter jobs (table + softmax jobs of the prologue / of the launch that
            # ends a forward) evaluated inside every step; a `cache_params` circuit evaluates none
            "params_recomputed_every_step": bool(not circuit.cache_params and circuit._batch is not None and len(circuit._batch) > 0),
            "param_jobs_per_step": (0 if circuit.cache_params or circuit._batch is None else len(circuit._batch)),
            "params_evaluated": ("at the start of each forward, by a launch of their own" if not getattr(circuit._bind(B), "params_at_end", False)
                                 else "once per forward, by the launch that ends it (for the next forward; a store that changed in "
                                      "between -- TensorStore.state() -- is re-evaluated at the start)"),
            "launches_per_step": int(circuit.num_launches_ll(B)),  # incl. the staging launch of the batch, if any
            "leaf_reads_raw_batch": bool(circuit.reads_batch_directly(B)),
        },
        "timing": {
            "rounds": len(walls), "steps_per_round": args.steps, "reported": "median round", "settle_steps": settle_steps,
            "ms_per_step_by_round": [1e3 * w / args.steps for w in walls],
            "hip_event_ms_per_step_by_round": evms,
            "input_batches_rotated": nb, "input_bytes_resident": nb * B * plan.num_variables * 8,
        },
        "distributed": {"backend": ("rccl-capi" if comm is not None else backend) if use_dist else None, "world_size": world,
                        "ranks_seen_by_backend": ranks_seen,
                        # rccl-capi: ck_comm_all_reduce_f64 on the launch stream behind every step (the library's own RCCL
                        # communicator); torch backends: one collective carries up to 64 steps' pairs
                        "every_step_exchanged": bool(use_dist),
                        "steps_per_collective": ((1 if comm is not None else 64) if use_dist else None),
                        "librccl": (comm.info()["librccl"] if comm is not None else None),
                        "rccl_capi_error": comm_error,
                        "collective_stream": (None if comm is None else ("launch stream" if args.sync_collectives
                                                                          else "the communicator's own stream, beside the next step")),
                        # a straggler GPU shows here: wall time per step of the fastest / slowest rank in the median round
                        "ms_per_step_fastest_rank": (1e3 * sorted(spread)[len(spread) // 2][0] / args.steps) if spread else None,
                        "ms_per_step_slowest_rank": (1e3 * sorted(spread, key=lambda t: t[1])[len(spread) // 2][1] / args.steps) if spread else None},
        "check": {"mean_ll": total_nll / max(total_rows, 1.0), "rows": total_rows},
    }

    # Secondary figures (never `value`).
    variants = {}
    if args.contraction == "f32" and not args.no_variants and world == 1:  # single-GPU extras only
        # "... and additionally reported cached": derived parameters (softmax, log
        # tables, tiled weights) kept from the previous step -- the serving configuration.
        alt = HipCircuit(plan, tensors, device=device, use_graph=not args.no_graph, fuse=fuse, cache_params=True)
        w3s, _, pair3 = timed_region(alt, args.steps, args.warmup, 3)
        w3 = float(np.median(w3s))
        variants["cache_params=True"] = {
            "what": "parameter graphs evaluated once and reused while the parameters do not change "
                    "(the reference, and `value`, recompute them inside every step)",
            "value": world * B * args.steps / w3,
            "ms_per_step": 1e3 * w3 / args.steps,
            "mean_ll": float(pair3[0]) / max(float(pair3[1]), 1.0),
        }
        del alt
        if not args.params_at_start:
            alt = HipCircuit(plan, tensors, device=device, use_graph=not args.no_graph, fuse=fuse, params_at_end=False)
            w5s, _, pair5 = timed_region(alt, args.steps, args.warmup, 3)
            w5 = float(np.median(w5s))
            variants["params_at_end=False"] = {
                "what": "the parameter graphs evaluated by a launch of their own at the START of every forward (3 launches per step) "
                        "instead of by the launch that ends the forward before (`value`: tail of forward k + parameters for forward "
                        "k+1 in one launch, ck_tail_params_fwd).  Every parameter graph is evaluated once per step, exact fp32, either way",
                "value": world * B * args.steps / w5,
                "ms_per_step": 1e3 * w5 / args.steps,
                "mean_ll": float(pair5[0]) / max(float(pair5[1]), 1.0),
                "launches_per_step": alt.num_launches_ll(B) if hasattr(alt, "num_launches_ll") else None,
            }
            del alt
        # The contraction off the fp32 lanes: every fp32 operand cut into bf16 pieces, products on the bf16
        # matrix pipe with fp32 accumulation.  Labelled variants: `value`, `dtype` and `roofline` stay exact fp32.
        gold = os.path.join(ROOT, "tests", "golden", "cfg2_qt784_golden.npz")
        for cname in ("bf16x3", "bf16x6"):
            alt = HipCircuit(plan, tensors, device=device, use_graph=not args.no_graph, fuse=fuse, contraction=cname)
            try:
                wbs, _, pairb = timed_region(alt, args.steps, args.warmup, 3)
            except ValueError as e:  # (not the depth-4 launch over the raw batch at this batch size)
                variants[f"contraction={cname}"] = {"skipped": str(e)}
                del alt
                continue
            wb = float(np.median(wbs))
            entry = {
                "what": ("the leaf launch's contractions on v_mfma_f32_32x32x16_bf16: operands split by truncation into "
                         + ("2 bf16 pieces, 3 products (~2^-15 per product)" if cname == "bf16x3" else "3 bf16 pieces, 6 products (fp32-like)")
                         + ", fp32 accumulation; every other launch exact fp32.  Not the reference's arithmetic: never `value`"),
                "value": world * B * args.steps / wb,
                "ms_per_step": 1e3 * wb / args.steps,
                "mean_ll": float(pairb[0]) / max(float(pairb[1]), 1.0),
            }
            if os.path.exists(gold):
                with np.load(gold) as z:
                    xg = torch.from_numpy(z["x"].astype(np.int64))
                    y64 = torch.from_numpy(z["y_f64"]).reshape(-1).double()
                    y32 = torch.from_numpy(z["y_f32"]).reshape(-1).double()
                xq = x.clone()
                xq[: xg.shape[0]] = xg.to(device)
                with torch.cuda.stream(stream):
                    ya = alt(xq).reshape(-1).double().cpu()[: xg.shape[0]]
                    yf = circuit(xq).reshape(-1).double().cpu()[: xg.shape[0]]
                entry["max_rel_err_vs_reference_fp64"] = float(((ya - y64).abs() / y64.abs()).max())
                entry["exact_f32_path_same_rows"] = float(((yf - y64).abs() / y64.abs()).max())
                entry["reference_fp32_same_rows"] = float(((y32 - y64).abs() / y64.abs()).max())
            variants[f"contraction={cname}"] = entry
            del alt
        # Two forwards in flight (cirkit_amd.circuit.HipCircuitStreams): the small latency-bound kernels of
        # one step (parameter prologue, fused tail) fill the bubbles of the other step's leaf kernel.
        from cirkit_amd.circuit import HipCircuitStreams

        pool = HipCircuitStreams(plan, circuit.store, n=2, device=device, wait_for_input=False,
                                 use_graph=not args.no_graph, fuse=fuse)  # the inputs are resident
        with torch.cuda.stream(stream):
            for k in range(max(args.warmup, 4)):
                pool.log_likelihood_sum(xs[k % nb])
        pool.synchronize()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for k in range(args.steps):
                pool.log_likelihood_sum(xs[k % nb])
        torch.cuda.synchronize(device)
        w4 = time.perf_counter() - t0
        del pool
        variants["streams=2"] = {
            "what": "steps issued alternately on two HIP streams (HipCircuitStreams: two circuits sharing the raw "
                    "parameters): consecutive forwards overlap on the device; per-step latency is unchanged",
            "value": world * B * args.steps / w4,
            "ms_per_step": 1e3 * w4 / args.steps,
        }

    if rank == 0:
        fwd_ms = step_ms_events
        n_cu = int(torch.cuda.get_device_properties(device).multi_processor_count)
        roof: dict = {
            "forward": {
                "what": "the whole step against HBM by ALGORITHMIC bytes (SURVEY.md 8d: every folded layer of the reference reads its "
                        "inputs and writes its output once); cross-layer fusion removes most of that traffic, so this exceeds 1 -- "
                        "it is the north-star's >= 0.30 figure, not a statement about any kernel",
                "algorithmic_bytes": alg["total"],
                "avg_ms": fwd_ms,
                "achieved_GBps": alg["total"] / (fwd_ms * 1e-3) / 1e9,
                "frac_of_8TBps": alg["total"] / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
        }
        pmc, pmc_source = None, None
        # (at N > 1 too: rank 0 measures its own device after the timed region while the other ranks wait at the closing
        #  barrier -- the N > 1 line carries the same `roofline` and `cpu_baseline` as the N = 1 line)
        if not args.no_live_pmc and local_rank == 0:
            child = ["--batch", str(B), "--fuse", str(args.fuse), "--contraction", args.contraction, "--batches", str(nb)]
            if args.no_graph:
                child.append("--no-graph")
            if args.staged_input:
                child.append("--staged-input")
            if args.params_at_start:
                child.append("--params-at-start")
            pmc = live_pmc(child)
            pmc_source = "three rocprofv3 counter passes run by this bench.py invocation (FETCH_SIZE x2, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES)"
        if pmc is None:
            tj = os.path.join(ROOT, "profiles", "traffic.json")
            # the committed passes of the default configuration (scripts/profile_round.sh r03 e); kernels are matched by label
            default_cfg = ([g.depth for g in circuit._groups] == [4] and len(circuit._tail) == 6 and args.contraction == "f32" and B == 4096
                           and not (args.staged_input or args.params_at_start))
            if os.path.exists(tj) and default_cfg:
                with open(tj, encoding="utf-8") as f:
                    tr = json.load(f)
                keys = sorted(k for k in tr if k[:1] == "r" and "," in k and k.split(",")[0][1:].isdigit() and len(k.split(",")) == 2)
                if keys:  # the latest committed passes of the default configuration ("r<round>,<tag>")
                    pmc, pmc_source = tr[keys[-1]], (f"profiles/traffic.json[{keys[-1]!r}] (committed rocprofv3 passes of the same "
                                                     "configuration; not measured in this run)")
        if not args.no_kernel_breakdown:
            with torch.cuda.stream(stream):
                rows = circuit.profile_kernels(x, iters=10)
            # aggregate per kernel
            agg: dict[str, dict] = {}
            for r in rows:
                a = agg.setdefault(r["kernel"], {"ms": 0.0, "bytes": 0.0, "flops": 0.0, "exec": 0.0, "launches": 0})
                a["ms"] += r["ms"]
                a["bytes"] += r["algorithmic_bytes"]
                a["flops"] += r.get("algorithmic_flops", 0.0)
                a["exec"] += r.get("executed_flops", 0.0)
                a["launches"] += 1
            params_bytes = float(sum(int(np.prod(shp)) * 4 for shp, _ in plan.tensors.values()))

            def kernel_fracs(k: str, v: dict) -> dict:
                """Per launch of kernel k: measured HBM bytes and MFMA-busy cycles against the two peaks (over the trace-timed
                duration where there is one), and the bytes the launch cannot avoid (what it reads that nobody on the chip holds
                + what it must write for the next launch)."""
                kp2 = next((pv for pk, pv in (pmc or {}).items() if pk.split("<")[0] == k.split("<")[0]), None)
                t = (kp2["trace_us"] * 1e-6) if (kp2 and kp2.get("trace_us")) else (v["ms"] * 1e-3 / v["launches"])
                out = {
                    "frac_hbm": (kp2["hbm_bytes"] / t / 1e9 / HBM_PEAK_GBS) if (kp2 and "hbm_bytes" in kp2) else None,
                    "frac_mfma": (kp2["mfma_busy_cycles"] / 64 * 4096 / t / 1e12 / FP32_MFMA_PEAK_TF) if (kp2 and "mfma_busy_cycles" in kp2) else None,
                }
                g0 = circuit._groups[0] if circuit._groups else None
                if g0 is not None and circuit.layers[g0.input_layer].__class__.__name__ == "HipCategoricalLayer":
                    cat = circuit.layers[g0.input_layer]
                    table = float(circuit.layers[g0.dense_layer].num_folds * (cat.num_categories + 1) * 32 * 4) if g0.dense_layer is not None else 0.0
                    roots = float(circuit.layers[g0.root].num_folds * B * 32 * 4)
                    if k.startswith("leaf_persistent_kernel"):  # batch + table in, root tiles out
                        out["compulsory_bytes"] = B * plan.num_variables * 8 + table + roots
                    elif k.startswith("tail_params_kernel"):  # root tiles + raw parameters in, table (+ weights) out
                        out["compulsory_bytes"] = roots + params_bytes + table + B * 4
                    elif k.startswith("softmax_batch_kernel"):
                        out["compulsory_bytes"] = params_bytes + table
                return out

            name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
            t_launch = a["ms"] * 1e-3 / a["launches"]
            kp = (pmc or {}).get(name.split("<")[0] if name not in (pmc or {}) else name) or (pmc or {}).get(name)
            if kp is None and pmc:  # template arguments may be printed differently by the profiler
                kp = next((v for k, v in pmc.items() if k.split("<")[0] == name.split("<")[0]), None)
            exec_tf = a["exec"] / a["launches"] / t_launch / 1e12
            hbm_frac = (kp["hbm_bytes"] / t_launch / 1e9 / HBM_PEAK_GBS) if kp and "hbm_bytes" in kp else None
            mfma_bound = a["exec"] > 0 and (hbm_frac is None or exec_tf / FP32_MFMA_PEAK_TF >= hbm_frac)
            roof.update({
                "kernel": name,
                "launches_per_step": a["launches"],
                "avg_us_per_launch": 1e6 * t_launch,
                "timed_with": "HIP events around every launch on the launch stream (instrumented eager pass of 10 forwards)",
                # the roofline that binds the dominant kernel: EXECUTED contraction flops (the MFMAs the launch issues; a
                # dense layer pushed through its category table is executed by the prologue and credited there) against
                # the dense fp32-input matrix peak, or measured HBM bytes against 8 TB/s, whichever fraction is larger
                "bound": "mfma" if mfma_bound else "hbm",
                "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "peak": FP32_MFMA_PEAK_TF if mfma_bound else HBM_PEAK_GBS,
                "achieved": exec_tf if mfma_bound else (kp["hbm_bytes"] / t_launch / 1e9 if kp and "hbm_bytes" in kp else None),
                # `frac`: with the launch duration of a plain rocprofv3 kernel trace (no events between the launches) when this run
                # has one -- what `profiles/` reproduces --, else with the event-timed duration; the latter always as
                # `frac_event_timed` (events around every launch cost each launch a few microseconds)
                "frac": ((a["exec"] / a["launches"] / (kp["trace_us"] * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF)
                         if (kp and kp.get("trace_us") and mfma_bound) else (exec_tf / FP32_MFMA_PEAK_TF if mfma_bound else hbm_frac)),
                "frac_timed_with": ("rocprofv3 kernel trace" if (kp and kp.get("trace_us") and mfma_bound) else "HIP events"),
                "frac_event_timed": exec_tf / FP32_MFMA_PEAK_TF if mfma_bound else hbm_frac,
                "trace_us_per_launch": kp.get("trace_us") if kp else None,
                "frac_trace_timed": (a["exec"] / a["launches"] / (kp["trace_us"] * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF)
                if (kp and kp.get("trace_us") and mfma_bound) else None,
                "executed_flops_per_launch": a["exec"] / a["launches"],
                "traffic": kp.get("hbm_bytes") if kp else None,
                # the counters as rocprofv3 reports them (FETCH_SIZE + WRITE_SIZE, KiB -> bytes) and the correction applied above
                "traffic_raw": kp.get("hbm_bytes_raw") if kp else None,
                "traffic_correction": "2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies a 128-byte read request at 64 bytes; the factor was "
                                      "measured as 2.000 for coalesced 16- and 4-byte-per-lane streams and for random 128- / 256-byte row "
                                      "gathers through registers and through LDS DMA (scripts/ubench/fetch_calib.hip, profiles/r05_fetch_calib.txt)",
                "traffic_source": pmc_source,
                "hbm_measured_frac": hbm_frac,
                "mfma_busy_frac": (kp["mfma_busy_cycles"] / (4 * n_cu * t_launch * 2.4e9)) if kp and "mfma_busy_cycles" in kp else None,
                "mfma_busy_frac_note": "SQ_VALU_MFMA_BUSY_CYCLES per launch / (4 SIMDs x CUs x launch time x 2.4 GHz)",
                # HBM bytes of the whole step (sum over its kernels, PMC) against what a step must move:
                # the raw parameters read once + the int64 batch read once + the per-row results
                "step_traffic": (sum(v.get("hbm_bytes", 0.0) for k2, v in (pmc or {}).items()
                                     if any(k2.split("<")[0] == kk.split("<")[0] for kk in agg)) or None) if pmc else None,
                "compulsory_bytes": float(sum(int(np.prod(shp)) * 4 for shp, _ in plan.tensors.values())
                                          + B * plan.num_variables * 8 + B * 4),
                "algorithmic": {
                    "what": "SURVEY.md 8d figures of the reference layers this launch stands for (not what it moves or executes)",
                    "bytes_per_launch": a["bytes"] / a["launches"], "flops_per_launch": a["flops"] / a["launches"],
                    "GBps": a["bytes"] / a["launches"] / t_launch / 1e9, "TFLOPps": a["flops"] / a["launches"] / t_launch / 1e12,
                },
                "kernels": {
                    k: {
                        "launches": v["launches"],
                        "ms_per_step": v["ms"],
                        "executed_TFLOPps": (v["exec"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else None,
                        "hbm_bytes_per_launch": next((pv.get("hbm_bytes") for pk, pv in (pmc or {}).items()
                                                      if pk.split("<")[0] == k.split("<")[0]), None),
                        "trace_us_per_launch": next((pv.get("trace_us") for pk, pv in (pmc or {}).items()
                                                     if pk.split("<")[0] == k.split("<")[0]), None),
                        **kernel_fracs(k, v),
                    }
                    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
                },
            })
        else:
            roof.update({"kernel": "forward program", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "achieved": roof["forward"]["achieved_GBps"], "frac": roof["forward"]["frac_of_8TBps"], "traffic": None})
        result["roofline"] = roof

        if not args.no_cpu_baseline:  # (rank 0, any N: timed on the host cores while the other ranks wait at the closing barrier)
            from oracle.torch_oracle import as_torch, evaluate_plan

            tt = as_torch(tensors)
            xc = x.cpu()
            # self-check of the headline path: the first 64 rows of batch 0 through the oracle (the reference's arithmetic)
            y_hip = circuit(x[:64]).reshape(-1).double().cpu()
            y_orc = evaluate_plan(plan, tt, xc[:64]).reshape(-1).double()
            result["check"]["max_rel_err_vs_oracle"] = float(((y_hip - y_orc).abs() / y_orc.abs()).max())
            result["check"]["rows_checked_against_oracle"] = 64
            # pick the host thread count that serves this op mix best (ATen's small-op overheads
            # make "all cores" far from optimal), on a 512-row probe
            best_thr, best_rate = 1, 0.0
            ncpu = os.cpu_count() or 1
            for thr in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
                torch.set_num_threads(thr)
                evaluate_plan(plan, tt, xc[:512])
                t1 = time.perf_counter()
                evaluate_plan(plan, tt, xc[:512])
                rate = 512 / (time.perf_counter() - t1)
                if rate > best_rate:
                    best_thr, best_rate = thr, rate
            torch.set_num_threads(best_thr)
            evaluate_plan(plan, tt, xc[:256])  # warm-up
            reps, t_cpu = 0, 0.0
            while reps < 5 and t_cpu < 15.0:
                t1 = time.perf_counter()
                evaluate_plan(plan, tt, xc)
                t_cpu += time.perf_counter() - t1
                reps += 1
            result["cpu_baseline"] = {
                "value": reps * B / t_cpu,
                "unit": "evals/s",
                "cores": torch.get_num_threads(),
                "host_cores": ncpu,
                "cpu_model": _cpu_model(),
                "kind": "port",
                "sample": f"{reps} x one {B}-row batch of the same workload through oracle/torch_oracle.py "
                          "(op-for-op restatement of the reference's torch-CPU forward, fp32, no_grad); `cores` = the ATen thread "
                          "count that was fastest on a 512-row probe, `host_cores` = what the box has",
            }
        result["variants"] = variants
        if world == 1 and not args.no_other_configs:
            result["other_configs"] = other_configs(device, stream, B)
        result["summary"] = bench_summary(result)  # LAST key: every figure DESIGN.md section 8 quotes, within the line's last 1500 characters
        print(json.dumps(result), flush=True)

    if use_dist:
        dist.barrier()
        if comm is not None:
            torch.cuda.synchronize(device)
            comm.destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
