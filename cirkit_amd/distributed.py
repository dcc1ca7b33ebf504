"""Data-parallel evaluation: shard the batch, replicate the parameters, one all-reduce.

The reference has no distributed code at all; log-likelihood evaluations
are independent across the batch axis, so the multi-GPU path is: rank r of R evaluates rows
``[r*B/R, (r+1)*B/R)`` with its own `HipCircuit` (one process per GPU), reduces them on device to
``[sum_b log p(x_b), count]`` (fp64, `ck_ll_sum`) and ONE ``all_reduce(SUM)`` of those 16 bytes over
RCCL/xGMI gives the global summed log-likelihood (SURVEY.md section 8 e).  No activation ever
crosses a GPU boundary; the collective is latency-bound, so bucket size / ring order are moot.

The helpers are backend-agnostic (`nccl` = RCCL on ROCm, `gloo` for the CPU tests).
"""

from __future__ import annotations

import os
from typing import Callable

import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced shard of ``n_rows`` (first ``n_rows % world`` ranks get one more)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_from_env(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    ``python -m torch.distributed.run``).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def all_reduce_ll(pair: torch.Tensor) -> torch.Tensor:
    """In-place SUM all-reduce of the ``[sum, count]`` pair (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(pair, op=dist.ReduceOp.SUM)
    return pair


class DataParallelEvaluator:
    """Wraps a per-rank ``ll_sum_fn(x_local) -> tensor([sum, count])`` (e.g.
    ``HipCircuit.log_likelihood_sum``) into a global mean-NLL evaluation."""

    def __init__(self, ll_sum_fn: Callable[[torch.Tensor], torch.Tensor]):
        self.ll_sum_fn = ll_sum_fn
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def local_rows(self, x_global: torch.Tensor) -> torch.Tensor:
        a, b = shard_bounds(x_global.shape[0], self.rank, self.world)
        return x_global[a:b]

    def summed_ll(self, x_local: torch.Tensor) -> torch.Tensor:
        """Global ``[sum log p, count]`` given this rank's shard (may be empty)."""
        if x_local.shape[0] == 0:
            pair = torch.zeros(2, dtype=torch.float64, device=x_local.device)
        else:
            pair = self.ll_sum_fn(x_local).to(torch.float64).clone()
        return all_reduce_ll(pair)

    def mean_nll(self, x_local: torch.Tensor) -> float:
        s = self.summed_ll(x_local)
        return -float(s[0]) / max(float(s[1]), 1.0)
