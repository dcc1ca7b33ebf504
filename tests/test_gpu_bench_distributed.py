"""`bench.py --gpus 2` launched exactly as the driver launches it (python -m torch.distributed.run, one process per rank),
on ONE device with the gloo backend (RCCL refuses two ranks on one GPU; BENCH_DIST_BACKEND=gloo exists for this): the
N > 1 path of the benchmark -- per-rank seeds, the ring of asynchronous all-reduces of the [sum, count] pair, max-over-ranks
timing, rank 0 printing ONE JSON line -- must report the global batch and the sum over ranks."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_gloo_on_one_device(hip_device):
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    steps, warmup, rounds, nb, B = 3, 1, 2, 2, 4096
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup),
           "--rounds", str(rounds), "--batches", str(nb), "--no-variants", "--no-other-configs", "--no-cpu-baseline",
           "--no-kernel-breakdown", "--no-live-pmc"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * B and d["scaling"] == "weak"
    assert d["distributed"] == {"backend": "gloo", "world_size": 2, "ranks_seen_by_backend": 2}
    assert d["check"]["rows"] == 2 * B
    assert d["value"] > 0 and abs(d["value"] - 2 * B / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    # the reported mean LL is the all-reduced [sum, count] of the LAST step: the same batches evaluated in this process
    plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32,
                      sum_product_layer="cp", num_sum_units=32)
    hc = HipCircuit(plan, init_plan_tensors(plan), device=hip_device)
    last = (warmup + rounds * steps - 1) % nb
    tot = torch.zeros(2, dtype=torch.float64)
    for rank in range(2):
        g = torch.Generator().manual_seed(1234 + rank)
        xs = [torch.randint(0, 256, (B, 784), generator=g) for _ in range(nb)]
        tot += hc.log_likelihood_sum(xs[last].to(hip_device)).cpu()
    assert tot[1].item() == 2 * B
    assert abs(d["check"]["mean_ll"] - tot[0].item() / tot[1].item()) <= 1e-9 * abs(tot[0].item() / tot[1].item())
