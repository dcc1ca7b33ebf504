"""Data-parallel evaluation: shard the batch, replicate the parameters, one all-reduce.

The reference has no distributed code at all; log-likelihood evaluations
are independent across the batch axis, so the multi-GPU path is: rank r of R evaluates rows
``[r*B/R, (r+1)*B/R)`` with its own `HipCircuit` (one process per GPU), reduces them on device to
``[sum_b log p(x_b), count]`` (fp64, `ck_ll_sum`) and ONE ``all_reduce(SUM)`` of those 16 bytes over
RCCL/xGMI gives the global summed log-likelihood (SURVEY.md section 8 e).  No activation ever
crosses a GPU boundary; the collective is latency-bound, so bucket size / ring order are moot.

The exchange itself goes through the C ABI when the tensors live on a GPU: `HipComm` (`ck_comm_*`, include/cirkit_hip.h) is
an RCCL communicator owned by the library; the collective is enqueued on the launch stream by the library itself.
torch.distributed is used for ONE thing in that case -- handing rank 0's `ncclUniqueId` to the other ranks
(`HipComm.from_process_group`; `HipComm.from_file` needs no process group at all) -- and remains the fallback of the HOST logic
where there is no GPU (the `gloo` tests on CPU).
"""

from __future__ import annotations

import ctypes
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


class HipComm:
    """An RCCL communicator behind the C ABI (`ck_comm_init` / `ck_comm_all_reduce_f64|f32` / `ck_comm_destroy`): in-place SUM
    all-reduces of fp32 / fp64 device tensors, enqueued on the CURRENT stream of the communicator's device (or recorded into a
    `ck_program`).  One per process and device."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: torch.device | str | int):
        from . import _capi as capi

        self.device = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if self.device.type != "cuda":
            raise capi.HipExtensionError("HipComm needs a ROCm device (the CPU tests of the host logic use torch.distributed / gloo)")
        if len(unique_id) != 128:
            raise ValueError("an ncclUniqueId is 128 bytes")
        self.rank, self.world = int(rank), int(world)
        self._capi = capi
        self._h = ctypes.c_void_p()
        _bind_rccl(capi)
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        with torch.cuda.device(self.device):
            capi.call("ck_comm_init", ctypes.cast(buf, ctypes.c_void_p), self.rank, self.world,
                      self.device.index if self.device.index is not None else torch.cuda.current_device(), ctypes.byref(self._h))

    # ---------------------------------------------------------------- bootstrap: only the 128-byte id travels outside RCCL
    @staticmethod
    def new_unique_id() -> bytes:
        from . import _capi as capi

        _bind_rccl(capi)
        buf = ctypes.create_string_buffer(128)
        capi.call("ck_comm_unique_id", ctypes.cast(buf, ctypes.c_void_p))
        return buf.raw

    @classmethod
    def from_process_group(cls, device: torch.device | str | int, group=None) -> "HipComm":
        """Rank 0 draws the id, torch.distributed (any backend, gloo included) broadcasts those 128 bytes, every rank joins."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], rank, world, device)

    @classmethod
    def from_file(cls, path: str, rank: int, world: int, device: torch.device | str | int, timeout_s: float = 120.0) -> "HipComm":
        """Without any process group: rank 0 writes the id to `path` (atomically), the others wait for the file."""
        import time

        if rank == 0:
            uid = cls.new_unique_id()
            with open(path + ".tmp", "wb") as f:
                f.write(uid)
            os.replace(path + ".tmp", path)
        else:
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > timeout_s:
                    raise TimeoutError(f"no ncclUniqueId at {path} after {timeout_s} s")
                time.sleep(0.01)
            with open(path, "rb") as f:
                uid = f.read()
        return cls(uid, rank, world, device)

    # ---------------------------------------------------------------- the exchange
    def all_reduce(self, t: torch.Tensor, stream: int | None = None) -> torch.Tensor:
        """In-place SUM over the ranks; ordered on `stream` (default: the current stream of the tensor's device)."""
        if self._h.value is None:
            raise self._capi.HipExtensionError("HipComm: communicator destroyed")
        if t.device != self.device and not (t.device.type == "cuda" and t.device.index == self.device.index):
            raise ValueError(f"HipComm on {self.device}: tensor on {t.device}")
        if not t.is_contiguous() or t.dtype not in (torch.float32, torch.float64):
            raise ValueError("HipComm.all_reduce: a contiguous float32 / float64 tensor")
        if t.numel() == 0:
            return t
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
            self._capi.call("ck_comm_all_reduce_f64" if t.dtype == torch.float64 else "ck_comm_all_reduce_f32",
                            self._h, t.data_ptr(), t.numel(), s)
        return t

    def all_reduce_async(self, t: torch.Tensor, after_stream: int | None = None) -> torch.Tensor:
        """In-place SUM of a contiguous float64 tensor on the communicator's OWN stream, ordered behind what the current stream
        (or `after_stream`) holds so far; the current stream does not wait -- call `wait()` before anything reads `t`."""
        if self._h.value is None:
            raise self._capi.HipExtensionError("HipComm: communicator destroyed")
        if t.dtype != torch.float64 or not t.is_contiguous() or t.device.type != "cuda" or t.device.index != self.device.index:
            raise ValueError(f"HipComm.all_reduce_async: a contiguous float64 tensor on {self.device}")
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream if after_stream is None else after_stream
            self._capi.call("ck_comm_all_reduce_async_f64", self._h, t.data_ptr(), t.numel(), s)
        return t

    def wait(self, stream: int | None = None) -> None:
        """The current stream (or `stream`) waits, on the device, for every `all_reduce_async` issued so far."""
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
            self._capi.call("ck_comm_wait", self._h, s)

    def info(self) -> dict:
        out = (ctypes.c_int32 * 3)()
        origin = ctypes.create_string_buffer(256)
        self._capi.call("ck_comm_info", self._h, out, origin, 256)
        return {"rank": int(out[0]), "world": int(out[1]), "device": int(out[2]), "librccl": origin.value.decode()}

    def destroy(self) -> None:
        global _default_comm
        if self._h.value is not None:
            h, self._h = self._h, ctypes.c_void_p()
            if _default_comm is self:
                _default_comm = None
            self._capi.call("ck_comm_destroy", h)


def _bind_rccl(capi) -> None:
    """RCCL is bound at run time: PyTorch-ROCm's own copy (built against the HIP runtime this process already holds) first."""
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    capi.call("ck_comm_load", cand.encode() if os.path.exists(cand) else None)


_default_comm: HipComm | None = None


def set_default_comm(comm: HipComm | None) -> None:
    """The communicator `all_reduce_ll`, `HipCircuit.log_likelihood_sum(reduce=True)` and the trainers' `all_reduce_grads` use."""
    global _default_comm
    _default_comm = comm


def default_comm() -> HipComm | None:
    return _default_comm


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    """In-place SUM all-reduce over the data-parallel ranks: through the C ABI (RCCL, on the current stream) when a `HipComm`
    is set and `t` lives on its device; through torch.distributed otherwise (the CPU / gloo tests of the host logic; RCCL via
    torch when no communicator was created); the identity for a single process."""
    c = _default_comm
    if c is not None and t.device.type == "cuda" and t.dtype in (torch.float32, torch.float64) and t.is_contiguous():
        return c.all_reduce(t)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def world_size() -> int:
    c = _default_comm
    if c is not None:
        return c.world
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def all_reduce_ll(pair: torch.Tensor) -> torch.Tensor:
    """In-place SUM all-reduce of the ``[sum, count]`` pair (no-op for a single process)."""
    if world_size() > 1 or _default_comm is not None:
        all_reduce_sum(pair)
    return pair


class DataParallelEvaluator:
    """Wraps a per-rank ``ll_sum_fn(x_local) -> tensor([sum, count])`` (e.g.
    ``HipCircuit.log_likelihood_sum``) into a global mean-NLL evaluation."""

    def __init__(self, ll_sum_fn: Callable[[torch.Tensor], torch.Tensor]):
        self.ll_sum_fn = ll_sum_fn
        c = default_comm()
        self.rank = c.rank if c is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.world = c.world if c is not None else (dist.get_world_size() if dist.is_initialized() else 1)

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
