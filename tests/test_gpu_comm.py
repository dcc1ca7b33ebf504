"""The exchange step through the C ABI (`ck_comm_*`, include/cirkit_hip.h; `cirkit_amd.distributed.HipComm`): RCCL itself at
world size 1 -- all a 1-GPU box can run; RCCL refuses two ranks on one device -- eagerly, inside a recorded `ck_program`, and
behind `HipCircuit.log_likelihood_sum(reduce=True)`.  The multi-rank HOST logic is covered on CPU with gloo
(tests/test_distributed_cpu.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_case


@pytest.fixture()
def comm(hip_device):
    from cirkit_amd.distributed import HipComm, set_default_comm

    c = HipComm(HipComm.new_unique_id(), 0, 1, hip_device)
    yield c
    torch.cuda.synchronize()
    set_default_comm(None)
    c.destroy()


@pytest.mark.gpu
def test_all_reduce_through_the_c_abi_at_world_size_one(hip_device, comm):
    info = comm.info()
    assert info["rank"] == 0 and info["world"] == 1 and "rccl" in info["librccl"]
    a = torch.arange(1000, dtype=torch.float64, device=hip_device)
    b = torch.linspace(-1, 1, 257, dtype=torch.float32, device=hip_device)
    a0, b0 = a.clone(), b.clone()
    s = torch.cuda.Stream(hip_device)
    with torch.cuda.stream(s):
        comm.all_reduce(a)
        comm.all_reduce(b)
    s.synchronize()
    assert torch.equal(a, a0) and torch.equal(b, b0)  # SUM over one rank
    # ... and on the communicator's own stream, ordered behind the work of the current one; `wait` orders readers behind it
    c = torch.zeros(6, dtype=torch.float64, device=hip_device)
    with torch.cuda.stream(s):
        c.fill_(2.5)
        comm.all_reduce_async(c)
        comm.wait()
        d = c * 2
    s.synchronize()
    assert d.tolist() == [5.0] * 6
    with pytest.raises(ValueError):
        comm.all_reduce_async(b)  # (float64 only)
    with pytest.raises(ValueError):
        comm.all_reduce(torch.zeros(4, dtype=torch.int32, device=hip_device))
    with pytest.raises(ValueError):
        comm.all_reduce(torch.zeros((4, 4), dtype=torch.float32, device=hip_device).t())


@pytest.mark.gpu
def test_all_reduce_is_a_step_of_a_recorded_program(hip_device, comm):
    """Inside ck_program_begin / _end the collective is appended to the launch list like any kernel of the library."""
    from cirkit_amd import _capi as capi

    buf = torch.full((2,), 3.0, dtype=torch.float64, device=hip_device)
    prog = C.c_void_p()
    capi.call("ck_program_begin", C.byref(prog))
    comm.all_reduce(buf)
    capi.call("ck_program_end", prog)
    assert capi.load().ck_program_num_ops(prog) == 1
    for _ in range(3):
        capi.call("ck_program_launch", prog, 0, torch.cuda.current_stream(hip_device).cuda_stream)
    torch.cuda.synchronize()
    assert buf.tolist() == [3.0, 3.0]
    capi.call("ck_program_destroy", prog)


@pytest.mark.gpu
def test_log_likelihood_sum_reduces_through_the_default_communicator(hip_device, comm, tmp_path):
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.distributed import DataParallelEvaluator, HipComm, all_reduce_ll, set_default_comm, world_size

    plan, tensors, g = load_case("cfg2_qt784")
    x = torch.from_numpy(g["x"].astype(np.int64)).to(hip_device)
    hc = HipCircuit(plan, tensors, device=hip_device)
    want = hc.log_likelihood_sum(x).clone()
    set_default_comm(comm)
    assert world_size() == 1
    got = hc.log_likelihood_sum(x, reduce=True).clone()
    out = torch.zeros(2, dtype=torch.float64, device=hip_device)
    hc.log_likelihood_sum(x, out=out, reduce=True)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(out, want)
    assert torch.equal(all_reduce_ll(want.clone()), want)
    ev = DataParallelEvaluator(hc.log_likelihood_sum)
    assert ev.world == 1 and abs(ev.mean_nll(x) + float(want[0] / want[1])) <= 1e-12 * abs(float(want[0]))
    # the bootstrap without any process group: the id through a file
    set_default_comm(None)
    c2 = HipComm.from_file(str(tmp_path / "id"), 0, 1, hip_device)
    t = torch.ones(5, dtype=torch.float32, device=hip_device)
    c2.all_reduce(t)
    torch.cuda.synchronize()
    assert t.sum().item() == 5.0
    c2.destroy()
    with pytest.raises(Exception):
        c2.all_reduce(t)
