"""The N > 1 path on CPU: two gloo processes, batch sharding + the single all-reduce.  The per-rank
evaluator is the CPU oracle here (no GPU in this container); what is under test is the host logic
of cirkit_amd/distributed.py that bench.py and the multi-GPU runs use."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, load_case
from cirkit_amd.distributed import shard_bounds


def test_shard_bounds_partition_the_batch():
    for n in (0, 1, 7, 8, 4096, 32768 + 5):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(32768, 3, 8) == (3 * 4096, 4 * 4096)  # BASELINE configs[2]
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _worker(rank, world, port, n_rows, out_path):
    import sys

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from conftest import load_case as lc
    from cirkit_amd.distributed import DataParallelEvaluator, init_from_env
    from oracle.torch_oracle import as_torch, evaluate_plan

    torch.set_num_threads(1)
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    plan, tensors, _ = lc("cfg1_rbt8")
    tt = as_torch(tensors)
    g = torch.Generator().manual_seed(7)
    x = torch.randint(0, 4, (n_rows, 8), generator=g)  # same global batch on every rank

    def ll_sum(xl):
        y = evaluate_plan(plan, tt, xl).double()
        return torch.stack([y.sum(), torch.tensor(float(xl.shape[0]), dtype=torch.float64)])

    from cirkit_amd.distributed import all_reduce_sum, default_comm, world_size

    # no GPU here, hence no HipComm: the host logic falls back to torch.distributed (gloo) for the same calls
    assert default_comm() is None and world_size() == world
    probe = all_reduce_sum(torch.tensor([1.0, float(rank)], dtype=torch.float32))
    assert probe.tolist() == [float(world), float(sum(range(world)))]
    ev = DataParallelEvaluator(ll_sum)
    s = ev.summed_ll(ev.local_rows(x))
    if rank == 0:
        np.save(out_path, s.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [64, 33, 1])
def test_two_process_gloo_all_reduce_matches_single_process(tmp_path, n_rows):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "pair.npy")
    mp.spawn(_worker, args=(2, port, n_rows, out), nprocs=2, join=True)
    pair = np.load(out)
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, _ = load_case("cfg1_rbt8")
    g = torch.Generator().manual_seed(7)
    x = torch.randint(0, 4, (n_rows, 8), generator=g)
    y = evaluate_plan(plan, as_torch(tensors), x).double()
    assert pair[1] == n_rows
    assert abs(pair[0] - float(y.sum())) <= 1e-6 * abs(float(y.sum()))  # fp32 evals, different GEMM blocking per shard


def test_hip_comm_needs_a_device_and_never_falls_back_silently():
    """`HipComm` is the GPU exchange (RCCL through the C ABI); on a CPU device it refuses instead of standing in gloo."""
    from cirkit_amd import _capi as capi
    from cirkit_amd.distributed import HipComm, all_reduce_sum, default_comm, world_size

    with pytest.raises(capi.HipExtensionError):
        HipComm(bytes(128), 0, 1, "cpu")
    assert default_comm() is None and world_size() == 1
    t = torch.arange(4, dtype=torch.float64)
    assert torch.equal(all_reduce_sum(t.clone()), t)  # a single process without a group: the identity
