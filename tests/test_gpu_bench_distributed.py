"""The N > 1 path of `bench.py` on a 1-GPU box.

* `bench.py --gpus 2` launched exactly as the driver launches it (python -m torch.distributed.run, one process per rank)
  AND as a plain `python bench.py --gpus 2` (bench.py then becomes the launcher itself), on ONE device with the gloo backend
  (RCCL refuses two ranks on one GPU; BENCH_DIST_BACKEND=gloo exists for this): per-rank seeds, the ring of asynchronous
  all-reduces of the [sum, count] pair, max-over-ranks timing, rank 0 printing ONE JSON line -- must report the global
  batch and the sum over ranks.
* RCCL itself at world size 1 (`bench.py --gpus 1 --dist`, backend "nccl" = librccl): the same code path -- process group
  bound to the device, ring buffers, `copy_`, `all_reduce(async_op=True)`, `work.wait()` on the launch stream, barriers,
  MAX-reduce of the wall time -- and `HipTrainer.all_reduce_grads` through the library; results equal the
  single-process ones bit for bit."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

STEPS, WARMUP, ROUNDS, NB, B = 3, 1, 2, 2, 4096
BENCH_ARGS = ["--steps", str(STEPS), "--warmup", str(WARMUP), "--settle", "0", "--rounds", str(ROUNDS), "--batches", str(NB), "--no-variants",
              "--no-other-configs", "--no-cpu-baseline", "--no-kernel-breakdown", "--no-live-pmc"]


def _one_json_line(out) -> dict:
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 only
    return json.loads(lines[0])


def _expected_last_step(hip_device, world: int) -> torch.Tensor:
    """[sum, count] over ranks of the LAST timed step, evaluated in this process from the per-rank seeds of bench.py."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32,
                      sum_product_layer="cp", num_sum_units=32)
    hc = HipCircuit(plan, init_plan_tensors(plan), device=hip_device)
    last = (WARMUP + ROUNDS * STEPS - 1) % NB
    tot = torch.zeros(2, dtype=torch.float64)
    for rank in range(world):
        g = torch.Generator().manual_seed(1234 + rank)
        xs = [torch.randint(0, 256, (B, 784), generator=g) for _ in range(NB)]
        tot += hc.log_likelihood_sum(xs[last].to(hip_device)).cpu()
    return tot


def _dist_fields(d: dict) -> dict:
    return {k: d["distributed"][k] for k in ("backend", "world_size", "ranks_seen_by_backend", "every_step_exchanged", "steps_per_collective")}


def _check_ranks(d: dict, hip_device, world: int) -> None:
    assert d["n_gpus"] == world and d["config"]["global_batch"] == world * B and d["scaling"] == "weak"
    assert _dist_fields(d) == {"backend": "gloo", "world_size": world, "ranks_seen_by_backend": world, "every_step_exchanged": True,
                               "steps_per_collective": 64}
    # per-rank wall time of the median round: a straggler would show as a gap between the two
    lo, hi = d["distributed"]["ms_per_step_fastest_rank"], d["distributed"]["ms_per_step_slowest_rank"]
    # (ranks that SHARE one device: the slowest rank of a round can be well above the median round -- only sanity-bound)
    assert 0 < lo <= hi <= 4.0 * d["ms_per_step"]
    assert d["check"]["rows"] == world * B
    assert d["steps_timed_total"] == ROUNDS * STEPS
    assert d["value"] > 0 and abs(d["value"] - world * B / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["cold_first_round_ms_per_step"] > 0
    # the reported mean LL is the all-reduced [sum, count] of the LAST step: the same batches evaluated in this process
    tot = _expected_last_step(hip_device, world)
    assert tot[1].item() == world * B
    assert abs(d["check"]["mean_ll"] - tot[0].item() / tot[1].item()) <= 1e-9 * abs(tot[0].item() / tot[1].item())


def _check_two_ranks(d: dict, hip_device) -> None:
    _check_ranks(d, hip_device, 2)


def test_bench_two_ranks_gloo_on_one_device(hip_device):
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", *BENCH_ARGS]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    _check_two_ranks(_one_json_line(out), hip_device)


def test_bench_launches_its_own_ranks(hip_device):
    """`python bench.py --gpus 2` with no launcher around it and no WORLD_SIZE in the environment (how a user, and the
    driver's 1-GPU invocation, call it): bench.py re-executes itself under torch.distributed.run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *BENCH_ARGS]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    _check_two_ranks(_one_json_line(out), hip_device)


def test_bench_eight_ranks_is_baseline_config_3(hip_device):
    """`python bench.py --gpus 8`, self-launched, eight ranks on the ONE device of this box over gloo: the launch, rendezvous,
    per-rank seeds and batches, bucketed exchange of every step's [sum, count], max-over-ranks timing and the single JSON line
    of BASELINE config 3 (32768 rows = 8 x 4096) -- everything of the driver's 8-GPU run but xGMI itself.  No scaling number
    comes out of this (eight processes share one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # (the N > 1 line is as complete as the N = 1 line -- rank 0 times the CPU oracle, checks 64 rows against it
    #  and fills `roofline` from its own instrumented pass while the other ranks wait at the closing barrier)
    args = [a for a in BENCH_ARGS if a not in ("--no-cpu-baseline", "--no-kernel-breakdown")]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", *args]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    d = _one_json_line(out)
    assert d["config"]["global_batch"] == 32768
    _check_ranks(d, hip_device, 8)
    c, r = d["cpu_baseline"], d["roofline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert d["check"]["max_rel_err_vs_oracle"] <= 1e-5 and d["check"]["rows_checked_against_oracle"] == 64
    assert r["kernel"].startswith("leaf_persistent_kernel") and r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1.0
    assert any(k.startswith("tail_params_kernel") for k in r["kernels"])


def test_bench_two_ranks_with_live_counter_passes(hip_device):
    """Rank 0 of an N = 2 run (launched as the driver launches it) runs the rocprofv3 counter passes on a plain single-process
    child -- none of the launcher's rendezvous variables may reach it -- so `roofline.traffic` is measured at N > 1 too (or,
    where rocprofv3 is not usable, taken from the committed profiles/traffic.json; never missing)."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = [a for a in BENCH_ARGS if a not in ("--no-kernel-breakdown", "--no-live-pmc")]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29619", os.path.join(ROOT, "bench.py"), "--gpus", "2", *args]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    d = _one_json_line(out)
    _check_two_ranks(d, hip_device)
    r = d["roofline"]
    assert r["traffic"] is not None and r["traffic"] > 0 and r["traffic_source"]
    assert r["kernel"].startswith("leaf_persistent_kernel") and 0 < r["frac"] <= 1.0


def test_bench_runs_rccl_at_world_size_one(hip_device):
    """The distributed path of bench.py through librccl (backend "nccl") with ONE rank: every collective of the N > 1 path
    is executed by RCCL; the numbers are those of a single process."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "BENCH_DIST_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist", *BENCH_ARGS]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    d = _one_json_line(out)
    # the exchange is the library's own RCCL communicator (ck_comm_*): one all-reduce per step on the launch stream
    assert _dist_fields(d) == {"backend": "rccl-capi", "world_size": 1, "ranks_seen_by_backend": 1, "every_step_exchanged": True,
                               "steps_per_collective": 1}
    assert "rccl" in d["distributed"]["librccl"]
    assert d["n_gpus"] == 1 and d["check"]["rows"] == B
    tot = _expected_last_step(hip_device, 1)
    assert d["check"]["mean_ll"] == tot[0].item() / tot[1].item()  # bit for bit: SUM over one rank is the identity


_TRAINER_SCRIPT = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data
from cirkit_amd.training import HipTrainer

use_dist = sys.argv[2] == "1"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if sys.argv[2] == "2":  # the library's own communicator, no process group at all: the id never leaves this process
    from cirkit_amd.distributed import HipComm, set_default_comm
    comm = HipComm(HipComm.new_unique_id(), 0, 1, dev)
    set_default_comm(comm)
    probe = torch.ones(3, dtype=torch.float32, device=dev)
    comm.all_reduce(probe)
    assert probe.tolist() == [1.0, 1.0, 1.0] and "rccl" in comm.info()["librccl"]
if use_dist:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    probe = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(probe)
    assert probe.item() == 1.0 and dist.get_backend() == "nccl"
plan = image_data((1, 8, 8), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32,
                  sum_product_layer="cp", num_sum_units=32)
tr = HipTrainer(plan, init_plan_tensors(plan), device=dev, lr=0.01)
g = torch.Generator().manual_seed(7)
lls = []
for _ in range(3):
    x = torch.randint(0, 256, (256, 64), generator=g).to(dev)
    lls.append(tr.step(x).cpu().tolist())
torch.cuda.synchronize()
h = float(sum(float(np.abs(v).sum()) for v in tr.parameters().values()))
print("RESULT " + json.dumps({"lls": lls, "param_abs_sum": h, "dist": use_dist}))
if sys.argv[2] == "2":
    assert tr.circuit is not None
    comm.destroy()
if use_dist:
    dist.destroy_process_group()
"""


def test_trainer_gradient_all_reduce_through_rccl(hip_device, tmp_path):
    """`HipTrainer.step` (forward, backward, `all_reduce_grads`, optimizer) with a one-rank RCCL process group against the
    same three steps without any process group: the same log-likelihoods and parameters."""
    script = tmp_path / "rccl_trainer.py"
    script.write_text(_TRAINER_SCRIPT)
    res = []
    for flag in ("1", "0", "2"):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
        out = subprocess.run([sys.executable, str(script), ROOT, flag], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        res.append(json.loads(next(l for l in out.stdout.splitlines() if l.startswith("RESULT "))[7:]))
    assert res[0]["dist"] and not res[1]["dist"]
    # (the backward accumulates weight gradients with float atomics across batch tiles: two runs of the same three steps
    # agree to rounding, not bit for bit -- with or without a process group; the first step's LL precedes any update)
    assert res[0]["lls"][0] == res[1]["lls"][0]
    for a, b in zip(res[0]["lls"], res[1]["lls"]):
        assert a[1] == b[1] and abs(a[0] - b[0]) <= 1e-6 * abs(b[0])
    assert abs(res[0]["param_abs_sum"] - res[1]["param_abs_sum"]) <= 1e-6 * res[1]["param_abs_sum"]
    # ... and through ck_comm_all_reduce_f32 (HipComm as the default communicator, no torch.distributed)
    assert res[2]["lls"][0] == res[1]["lls"][0]
    for a, b in zip(res[2]["lls"], res[1]["lls"]):
        assert a[1] == b[1] and abs(a[0] - b[0]) <= 1e-6 * abs(b[0])
    assert abs(res[2]["param_abs_sum"] - res[1]["param_abs_sum"]) <= 1e-6 * res[1]["param_abs_sum"]


def test_bench_line_carries_the_contract(hip_device):
    """`python bench.py` (N = 1, short): ONE JSON line with the fields the driver reads -- metric / value / unit / n_gpus / steps /
    warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload -- plus `roofline` (dominant
    kernel: bound, achieved, peak, unit, frac, traffic) and `cpu_baseline` (value, unit, cores, kind, sample); value = rows / time."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "BENCH_DIST_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--settle", "20", "--rounds", "2",
           "--no-variants", "--no-other-configs", "--no-live-pmc"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _one_json_line(out)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "steps_timed_total", "first_round_ms_per_step"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "evals/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    assert r["kernel"].startswith("leaf_persistent_kernel")
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert d["check"]["max_rel_err_vs_oracle"] <= 1e-5
