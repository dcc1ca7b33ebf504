"""Complex-valued circuits on linear (re, im) tiles (`cirkit_amd/circuit_clin.py`, `csrc/ck_clin.hip`): BASELINE config 5's
c(x) evaluated as complex-valued parameters require -- real parameters forced onto the path, complex Embedding weights, complex
sum weights -- against the reference's golden output, the oracle in complex128 (the reference's arithmetic,
ComplexLSESumSemiring.apply_reduce semiring.py:441-476) and the layer-wise complex kernels (`complex_linear=False`); the
layer launch alone against a complex128 restatement."""
import numpy as np
import pytest
import torch

from conftest import load_case

pytestmark = pytest.mark.gpu
REL = 1e-4


def _x_of(g):
    return torch.from_numpy(g["x"].astype(np.int64))


def _fp64(plan, tensors, x):
    from oracle.torch_oracle import as_torch, evaluate_plan

    t64 = {k: (v.to(torch.complex128) if v.is_complex() else v.double()) for k, v in as_torch(tensors).items()}
    torch.set_default_dtype(torch.float64)
    try:
        return evaluate_plan(plan, t64, x)
    finally:
        torch.set_default_dtype(torch.float32)


def _phase_gap(a, b):
    return float((torch.exp(1j * a.double()) - torch.exp(1j * b.double())).abs().max())


@pytest.mark.parametrize("depth", [1, 2, 3, 4])
def test_config5_on_linear_tiles_matches_the_reference(hip_device, depth, monkeypatch):
    """Real parameters, `signed_real=False`: what bench.py times as `config5_complex_weights`.  Every leaf-launch depth."""
    from cirkit_amd.circuit import HipCircuit

    monkeypatch.setenv("CK_CLIN_DEPTH", str(depth))
    plan, tensors, g = load_case("cfg5_sos_c_k32")
    x = _x_of(g)
    hc = HipCircuit(plan, tensors, device=hip_device, signed_real=False)
    assert hc._clin is not None and hc._clin.D == depth and not hc._clin.table_complex
    y = hc(x.to(hip_device)).cpu()
    y2 = hc(x.to(hip_device)).cpu()  # (the recorded launch list replayed)
    assert torch.equal(y.real, y2.real)
    yr = torch.from_numpy(g["y_c64"])
    assert y.shape == yr.shape and y.dtype == torch.complex64
    assert float(((y.real - yr.real).abs() / yr.real.abs()).max()) <= REL
    assert _phase_gap(y.imag, yr.imag) <= 5e-3
    old = HipCircuit(plan, tensors, device=hip_device, signed_real=False, complex_linear=False)
    assert old._clin is None
    yo = old(x.to(hip_device)).cpu()
    assert float(((y.real - yo.real).abs() / yo.real.abs()).max()) <= REL and _phase_gap(y.imag, yo.imag) <= 5e-3
    # closer to complex128 than the reference's own complex64 run is (or within 1e-6 of the value)
    y64 = _fp64(plan, tensors, x)
    mine = float((y.real.double() - y64.real).abs().max())
    ref = float((yr.real.double() - y64.real).abs().max())
    assert mine <= 4.0 * ref + 1e-6 * float(y64.real.abs().max()), (mine, ref)


def test_leaf_launch_reads_and_validates_the_raw_batch(hip_device, monkeypatch):
    """The leaf launch can take the caller's (B, D) int64 tensor itself (`ck_clin_leaf_fwd` with x_input; `CK_CLIN_RAW=1`: measured
    slower than the staged copy, which stays the default): no staged copy, no poison launch -- the same bits as over the staged batch; a category outside the Embedding layer's range (what
    TorchEmbeddingLayer's indexing raises for, layers/input.py:258-266) makes ITS row NaN, leaves the others alone and raises the
    flag `check_inputs()` turns into IndexError; batches rotate without re-recording."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg5_sos_c_k32")
    gen = torch.Generator().manual_seed(12)
    xa = torch.randint(0, 256, (200, plan.num_variables), generator=gen).to(hip_device)
    xb = torch.randint(0, 256, (200, plan.num_variables), generator=gen).to(hip_device)
    b = HipCircuit(plan, tensors, device=hip_device, signed_real=False)  # (the default: a staged (D, B) int32 copy)
    monkeypatch.setenv("CK_CLIN_RAW", "1")
    a = HipCircuit(plan, tensors, device=hip_device, signed_real=False)
    assert a._bind(200).direct and not b._bind(200).direct
    assert a.num_launches(200) == b.num_launches(200) - 1  # (no poison launch in the recorded list; nor the staging launch in front of it)
    for x in (xa, xb, xa):
        ya, yb = a(x).cpu(), b(x).cpu()
        assert torch.equal(ya.real, yb.real) and torch.equal(ya.imag, yb.imag)
    a.check_inputs()
    bad = xa.clone()
    bad[7, 300] = 256
    bad[150, 2] = 1 << 40
    y = a(bad).cpu()
    nan_rows = torch.isnan(y.real.reshape(200, -1)).any(dim=1)
    assert nan_rows.nonzero().reshape(-1).tolist() == [7, 150]
    ok = ~nan_rows
    assert torch.equal(y.real.reshape(200, -1)[ok], a(xa).cpu().real.reshape(200, -1)[ok])
    with pytest.raises(IndexError):
        a.check_inputs()
    a.check_inputs()  # (the flag was consumed)
    quiet = HipCircuit(plan, tensors, device=hip_device, signed_real=False, validate_inputs=False)  # (built under CK_CLIN_RAW=1)
    assert bool(torch.isfinite(quiet(bad).real).all())  # (clamped to the last category, silently)


@pytest.mark.parametrize("complex_sums", [False, True])
def test_tail_launch_is_bit_identical_to_the_layer_launches(hip_device, complex_sums, monkeypatch):
    """The few-fold top of the circuit in one launch (`ck_clin_tail_fwd`: a workgroup per 32-row tile walks the layers) against
    one launch per layer: the same arithmetic in the same order, the same bits; 4100 rows: a ragged last tile."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg5_sos_c_k32")
    rng = np.random.default_rng(3)
    if complex_sums:
        tensors = {k: (np.asarray(v) * np.exp(1j * rng.uniform(-np.pi, np.pi, np.asarray(v).shape))).astype(np.complex64) for k, v in tensors.items()}
    x = torch.randint(0, 256, (4100, plan.num_variables), generator=torch.Generator().manual_seed(8)).to(hip_device)
    a = HipCircuit(plan, tensors, device=hip_device, signed_real=False)
    monkeypatch.setenv("CK_CLIN_TAIL", "0")
    b = HipCircuit(plan, tensors, device=hip_device, signed_real=False)
    assert a._clin is not None and len(a._clin.tail) >= 5 and b._clin is not None and not b._clin.tail
    assert a.num_launches(4100) < b.num_launches(4100)
    ya, yb = a(x).cpu(), b(x).cpu()
    assert torch.equal(ya.real, yb.real) and torch.equal(ya.imag, yb.imag) and bool(torch.isfinite(ya.real).all())


@pytest.mark.parametrize("complex_emb,complex_sums", [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize("rows", [64, 45])
def test_complex_parameters_on_linear_tiles(hip_device, complex_emb, complex_sums, rows):
    """Complex Embedding weights (a complex table), complex sum weights (four MFMA chains per contraction), both; a batch that
    is not a multiple of the 32-row tile.  Against the oracle in complex128, the oracle's complex64 error as yardstick."""
    from cirkit_amd.circuit import HipCircuit
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, g = load_case("cfg5_sos_c_k32")
    rng = np.random.default_rng(11)
    emb = {n.config["tensor"] for sp in plan.layers if sp.type == "embedding" for n in sp.params["weight"].nodes if n.op == "tensor"}
    t2 = {}
    for k, v in tensors.items():
        v = np.asarray(v)
        if (k in emb and complex_emb) or (k not in emb and complex_sums):
            v = (v * np.exp(1j * rng.uniform(-np.pi, np.pi, v.shape))).astype(np.complex64)
        t2[k] = v
    x = _x_of(g)[:rows]
    hc = HipCircuit(plan, t2, device=hip_device)
    assert hc._clin is not None and hc._clin.table_complex == complex_emb and all(v == complex_sums for v in hc._clin.wcx.values())
    y = hc(x.to(hip_device)).cpu()
    y64 = _fp64(plan, t2, x)
    y32 = evaluate_plan(plan, as_torch(t2), x)
    d_ref = float((y32.real.double() - y64.real).abs().max())
    assert torch.isfinite(y.real).all()
    assert float((y.real.double() - y64.real).abs().max()) <= 4.0 * d_ref + 1e-5 * float(y64.real.abs().max())
    assert _phase_gap(y.imag, y64.imag) <= 4.0 * _phase_gap(y32.imag, y64.imag) + 2e-3


@pytest.mark.parametrize("H,Ko,wc", [(2, 32, False), (1, 32, True), (2, 1, True), (3, 7, False)])
def test_layer_launch_against_complex128(hip_device, H, Ko, wc):
    """`ck_clin_layer_fwd` alone: F folds of H children on tile blocks with random exponents, Ko <= 32 outputs, written both as a
    tile block and as the reference's (log|v|, arg v) pairs."""
    import ctypes as C

    from cirkit_amd import _capi as capi

    g = torch.Generator().manual_seed(3)
    F, Fc, B = 5, 7, 77
    tiles = (B + 31) // 32
    val = torch.randn((Fc, tiles * 32, 32), generator=g, dtype=torch.float64) + 1j * torch.randn((Fc, tiles * 32, 32), generator=g, dtype=torch.float64)
    val[2, 5] = 0  # (a row of zeros: log -> -inf)
    ex = torch.randint(-40, 40, (Fc, tiles * 32), generator=g, dtype=torch.int32)
    # tile-native blocks: dword 256 g + 4 lane + t of a tile = unit 8 g + 4 (lane >> 5) + t of row lane & 31
    blk = torch.empty((Fc, tiles, 2, 1024), dtype=torch.float32)
    gg, ln, tt = torch.meshgrid(torch.arange(4), torch.arange(64), torch.arange(4), indexing="ij")
    unit, row, pos = (8 * gg + 4 * (ln >> 5) + tt).reshape(-1), (ln & 31).reshape(-1), (256 * gg + 4 * ln + tt).reshape(-1)
    v32 = val.to(torch.complex64)
    for t in range(tiles):
        blk[:, t, 0, pos] = v32.real[:, t * 32 + row, unit]
        blk[:, t, 1, pos] = v32.imag[:, t * 32 + row, unit]
    child = torch.randint(0, Fc, (F, H), generator=g)
    w = torch.randn((F, Ko, 32), generator=g, dtype=torch.float64) + (1j * torch.randn((F, Ko, 32), generator=g, dtype=torch.float64) if wc else 0)
    w32 = w.to(torch.complex64 if wc else torch.float32).real.contiguous() if not wc else w.to(torch.complex64).contiguous()
    dev = hip_device
    lin, lin_e, wd = blk.to(dev), ex.to(dev), w32.to(dev)
    co = (child * tiles * 2048).to(torch.int64).to(dev)
    ce = (child * tiles * 32).to(torch.int64).to(dev)
    per = Ko * 32 * (8 if wc else 4)
    wptr = torch.tensor([wd.data_ptr() + f * per for f in range(F)], dtype=torch.int64, device=dev)
    out = torch.zeros((F, tiles, 2, 1024), dtype=torch.float32, device=dev)
    out_e = torch.zeros((F, tiles * 32), dtype=torch.int32, device=dev)
    out_log = torch.zeros((F, B, Ko), dtype=torch.complex64, device=dev)
    capi.call("ck_clin_layer_fwd", lin.data_ptr(), lin_e.data_ptr(), co.data_ptr(), ce.data_ptr(), wptr.data_ptr(), 1 if wc else 0,
              out.data_ptr(), out_e.data_ptr(), out_log.data_ptr(), F, H, Ko, B, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    # complex128: product of the children's fp32 values (what the launch reads), times 2^(sum of exponents), through W
    v = v32.to(torch.complex128)
    want = torch.empty((F, B, Ko), dtype=torch.complex128)
    e_sum = torch.zeros((F, B), dtype=torch.float64)
    for f in range(F):
        p = torch.ones((B, 32), dtype=torch.complex128)
        for h in range(H):
            p = p * v[child[f, h], :B]
            e_sum[f] += ex[child[f, h], :B].double()
        want[f] = p @ (w32[f].to(torch.complex128)).T
    want_log = torch.log(want.abs()) + e_sum[..., None] * np.log(2.0)
    got = out_log.cpu()
    fin = torch.isfinite(want_log)
    assert torch.equal(torch.isfinite(got.real), fin)
    assert float((got.real.double()[fin] - want_log[fin]).abs().max()) <= 2e-5 * max(1.0, float(want_log[fin].abs().max()))
    strong = fin & (want.abs() > 1e-3 * want.abs().amax(dim=-1, keepdim=True))  # (phases of sums that did not cancel)
    assert _phase_gap(got.imag[strong], torch.angle(want)[strong]) <= 1e-3
    # the tile block holds the same numbers: re + i im times 2^e
    o, oe = out.cpu(), out_e.cpu()
    for t in range(tiles):
        rows = t * 32 + row
        live = rows < B
        zr = o[:, t, 0, pos][:, live].double()
        zi = o[:, t, 1, pos][:, live].double()
        u, rr = unit[live], rows[live]
        keep = u < Ko
        mag = torch.log(torch.hypot(zr, zi)) + oe[:, rr].double() * np.log(2.0)
        wl = want_log[:, rr, :][:, torch.arange(len(rr)), u.clamp_max(Ko - 1)]
        ok = keep[None, :] & torch.isfinite(wl)
        assert float((mag[ok] - wl[ok]).abs().max()) <= 2e-5 * max(1.0, float(wl[ok].abs().max()))
