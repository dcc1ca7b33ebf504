"""Plan-level training of a squared circuit (complex-lse-sum for c, Z from ConstantValue / Hadamard / TensorDot layers with
pointer / conj / einsum / flatten parameter graphs): `cirkit_amd.training_squared.HipSquaredTrainer` against the gradients of
the reference's own ``loss.backward()`` for ``loss = -mean(2 Re c(x) - Re Z)`` (tests/golden/sos_4x4_k4_grads.npz,
make_fixtures.py `grads_sos`)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.plan import Plan  # noqa: E402


def _case():
    plan_c, plan_z = Plan.load(os.path.join(GOLDEN, "sos_4x4_c_k4")), Plan.load(os.path.join(GOLDEN, "sos_4x4_z_k4"))
    with np.load(os.path.join(GOLDEN, "sos_4x4_k4_grads.npz")) as z:
        ref = {k: z[k] for k in z.files}
    return plan_c, plan_z, init_plan_tensors(plan_c), ref


def test_trainer_points_at_the_squared_trainer_for_complex_plans():
    from cirkit_amd.training import HipTrainer

    plan_c, _, tensors, _ = _case()
    with pytest.raises(NotImplementedError, match="HipSquaredTrainer"):
        HipTrainer(plan_c, tensors, device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("native_z", [False, True])
def test_squared_circuit_gradients_match_the_reference(hip_device, native_z):
    """Both circuits through the HIP forward, both reverse launch lists, the parameter graphs of Z in reverse mode: the
    loss and every parameter gradient of the reference's autograd.  native_z: Z built from the plan of c
    (cirkit_amd/functional.py) instead of the plan the reference compiled."""
    from cirkit_amd.training_squared import HipSquaredTrainer

    plan_c, plan_z, tensors, ref = _case()
    tr = HipSquaredTrainer(plan_c, tensors, plan_z=None if native_z else plan_z, device=hip_device)
    x = torch.from_numpy(ref["x"].astype(np.int64)).to(hip_device)
    ll = tr.loss_and_grads(x).cpu().numpy()
    loss = -ll[0] / ll[1]
    assert abs(loss - float(ref["loss"])) <= 2e-5 * abs(float(ref["loss"])), (loss, float(ref["loss"]))
    got = tr.gradients()
    for k in tensors:
        want = ref["g_" + k]
        err = float(np.abs(got[k] - want).max())
        assert err <= 1e-3 * max(1e-3, float(np.abs(want).max())), (k, err, float(np.abs(want).max()))


@pytest.mark.gpu
def test_squared_circuit_training_steps_increase_the_likelihood(hip_device):
    from cirkit_amd.training_squared import HipSquaredTrainer

    plan_c, _, tensors, ref = _case()
    tr = HipSquaredTrainer(plan_c, tensors, device=hip_device, lr=0.01)
    x = torch.from_numpy(ref["x"].astype(np.int64)).to(hip_device)
    lls = [float(tr.step(x)[0]) for _ in range(8)]
    assert all(np.isfinite(lls)) and lls[-1] > lls[0], lls
    # the parameters the two circuits share moved, and the normaliser follows them: sum_x |c(x)|^2 / Z stays a density
    assert any(np.abs(tr.parameters()[k] - tensors[k]).max() > 1e-4 for k in tensors)


@pytest.mark.gpu
@pytest.mark.parametrize("signed", [False, True, "layers"])
def test_squared_trainer_on_baseline_config_5_against_the_oracles_autograd(hip_device, signed, monkeypatch):
    """BASELINE config 5 (QuadTree 28x28, Embedding-256, CP-T, K = 32; Z from its own plan): the gradients of 32 rows against
    torch autograd through the oracle's restatement of the reference forward (bit-identical to the reference on CPU).
    signed: c(x) on signed-log blocks (ck_signed.hip: fp32 log|v| + a sign bit, Embedding rows gathered by the first sum
    layer) instead of the complex layer-wise launch list -- what the trainer picks by itself for this circuit, with the
    Embedding -> four CP-T levels region on signed LINEAR tiles (one forward launch with kept tiles, two backward launches:
    `ck_leaf_walk_fwd / _bwd` signed); "layers": every signed-log layer a launch of its own (`CK_SLSE_LEAF=0`)."""
    from cirkit_amd.training_squared import HipSquaredTrainer
    from oracle import torch_oracle as oracle  # (tests may: the oracle is the checker)

    if signed == "layers":
        monkeypatch.setenv("CK_SLSE_LEAF", "0")
        signed = True

    plan_c, plan_z = Plan.load(os.path.join(GOLDEN, "cfg5_sos_c_k32")), Plan.load(os.path.join(GOLDEN, "cfg5_sos_z_k32"))
    tensors = init_plan_tensors(plan_c)
    # (the counter-based generator emits a few exact zeros: log 0 under autograd is NaN in the reference as well)
    tensors = {k: np.where(v == 0, np.float32(1e-2), v).astype(np.float32) for k, v in tensors.items()}
    x = torch.randint(0, 256, (32, 784), generator=torch.Generator().manual_seed(5))
    leaves = {k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.float64).requires_grad_(True) for k, v in tensors.items()}
    c = oracle.evaluate_plan(plan_c, leaves, x, grad=True)
    z = oracle.evaluate_plan(plan_z, leaves, None, grad=True)
    loss = -(2.0 * c.real - z.real).mean()
    loss.backward()
    tr = HipSquaredTrainer(plan_c, tensors, plan_z=plan_z, device=hip_device, signed=signed)
    assert (tr._signed is not None) == signed
    if signed:
        leaf = tr._signed.leaf
        assert (leaf is None) == (os.environ.get("CK_SLSE_LEAF") == "0") and (leaf is None or leaf.depth == 4)
    assert HipSquaredTrainer(plan_c, tensors, plan_z=plan_z, device=hip_device)._signed is not None  # (the default for this circuit)
    for _ in range(4):  # (eager, eager, recorded, replayed: the same numbers every time)
        ll = tr.loss_and_grads(x.to(hip_device)).cpu().numpy()
        assert abs(-ll[0] / ll[1] - loss.item()) <= 1e-4 * abs(loss.item()), (-ll[0] / ll[1], loss.item())
        got = tr.gradients()
        for k in tensors:
            want = leaves[k].grad.numpy()
            err = float(np.abs(got[k] - want).max())
            assert err <= 2e-3 * max(1e-6, float(np.abs(want).max())), (k, err, float(np.abs(want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [300, 4096])
def test_leaf_region_on_linear_tiles_agrees_with_the_layer_launches(hip_device, rows, monkeypatch):
    """Config 5's c(x): Embedding -> four CP-T levels in ONE forward launch on signed linear tiles (kept tiles of levels 2 and
    4) and two backward launches, against one signed-log launch per layer (`CK_SLSE_LEAF=0`) -- other arithmetic (products
    renormalised by powers of two instead of exp / log per layer), so values agree to rounding: log|c(x)| per row, the
    log-likelihood, every gradient (against the layer-wise one, in the norm: a batch holds rows whose c(x) nearly cancels and
    amplifies rounding, LAB_NOTES R5.3); 300 rows: a ragged last tile; eager and recorded launches give the same numbers."""
    from cirkit_amd.training_squared import HipSquaredTrainer

    plan_c, plan_z = Plan.load(os.path.join(GOLDEN, "cfg5_sos_c_k32")), Plan.load(os.path.join(GOLDEN, "cfg5_sos_z_k32"))
    tensors = init_plan_tensors(plan_c)
    tensors = {k: np.where(v == 0, np.float32(1e-2), v).astype(np.float32) for k, v in tensors.items()}
    x = torch.randint(0, 256, (rows, 784), generator=torch.Generator().manual_seed(6)).to(hip_device)
    a = HipSquaredTrainer(plan_c, tensors, plan_z=plan_z, device=hip_device)
    monkeypatch.setenv("CK_SLSE_LEAF", "0")
    b = HipSquaredTrainer(plan_c, tensors, plan_z=plan_z, device=hip_device)
    assert a._signed.leaf is not None and a._signed.leaf.depth == 4 and b._signed.leaf is None
    if rows > 1000:
        # a batch this large holds rows whose c(x) nearly cancels: their log|c| differs between ANY two arithmetics (here by up to
        # 0.6) and, since every gradient of a row carries 1 / c(x_b), one such row outweighs the rest of the batch (LAB_NOTES R5.3).
        # Values are compared on all rows; gradients on the rows both paths agree on (ragged again: another tile count).
        a.loss_and_grads(x), b.loss_and_grads(x)
        dy = (a._signed.output(rows).cpu().double() - b._signed.output(rows).cpu().double()).abs()
        assert float(dy.median()) <= 5e-4 and float((dy > 1e-2).double().mean()) <= 0.01, (float(dy.median()), float(dy.max()))
        x = x[(dy <= 2e-4).to(x.device)][:3000].contiguous()
        rows = int(x.shape[0])
        assert rows >= 2000
    lb = b.loss_and_grads(x).cpu().numpy()
    yb, gb = b._signed.output(rows).cpu().double(), b.gradients()
    for it in range(4):  # (eager, eager, recorded, replayed)
        la = a.loss_and_grads(x).cpu().numpy()
        ya = a._signed.output(rows).cpu().double()
        assert bool(torch.isfinite(ya).all())
        dy = (ya - yb).abs()  # (log|c| ~ 1300: 1e-7 relative is 1e-4; a row whose c(x) nearly cancels shows the amplified rounding)
        assert float(dy.median()) <= 5e-4 and float((dy > 1e-2).double().mean()) <= 0.01, (it, float(dy.median()), float(dy.max()))
        assert la[1] == lb[1] == rows and abs(la[0] - lb[0]) <= 1e-5 * abs(lb[0]), (la, lb)
        ga = a.gradients()
        for k in tensors:
            d, n = float(np.linalg.norm((ga[k] - gb[k]).astype(np.float64))), float(np.linalg.norm(gb[k].astype(np.float64)))
            assert d <= 2e-2 * n + 1e-12, (it, k, d, n)
    assert int(a._signed.bind(rows)["leaf"]["redo"].sum()) == 0  # (no tile left the linear range)


@pytest.mark.gpu
def test_leaf_region_tiles_that_leave_the_linear_range(hip_device, monkeypatch):
    """Embedding rows whose large units do not overlap: the products of the first levels fall below the linear floor, the
    forward marks such (root, tile) units and evaluates them in signed log space (`leaf_signed_redo_kernel`), the backward
    launches skip their kept tiles and `ck_leaf_walk_bwd_redo` (is_signed) walks them -- same values and gradients as the
    layer-wise launch list, marks cleared."""
    from cirkit_amd.templates import image_data
    from cirkit_amd.training_squared import HipSquaredTrainer

    plan_c = image_data((1, 8, 8), "quad-tree-2", input_layer="embedding", num_input_units=32, sum_product_layer="cp-t", num_sum_units=32,
                        sum_weight_activation="none", semiring="complex-lse-sum")
    tensors = init_plan_tensors(plan_c, seed=3)
    emb = next(n.config["tensor"] for sp in plan_c.layers if sp.type == "embedding" for n in sp.params["weight"].nodes if n.op == "tensor")
    w = tensors[emb].copy()  # (F, 32, C)
    half = np.arange(32) < 16
    var = np.arange(w.shape[0]) % 2 == 0
    w[np.ix_(var, half, [0, 1])] *= 1e-20   # states 0 / 1 of even variables live on the upper units, of odd ones on the lower
    w[np.ix_(~var, ~half, [0, 1])] *= 1e-20
    tensors = {k: np.where(v == 0, np.float32(1e-2), v).astype(np.float32) for k, v in {**tensors, emb: w}.items()}
    B = 200
    x = torch.randint(0, 4, (B, 64), generator=torch.Generator().manual_seed(2)).to(hip_device)
    a = HipSquaredTrainer(plan_c, tensors, device=hip_device)
    monkeypatch.setenv("CK_SLSE_LEAF", "0")
    b = HipSquaredTrainer(plan_c, tensors, device=hip_device)
    assert a._signed is not None and a._signed.leaf is not None and b._signed.leaf is None
    stream = torch.cuda.current_stream().cuda_stream
    a._signed.stage(x, stream)
    a._signed.forward(B, stream)
    marks = a._signed.bind(B)["leaf"]["redo"]
    assert int(marks.sum()) > 0  # (the case this test is about)
    a._flat_grad.zero_()
    a._signed.backward(B, -2.0 / B, stream)
    torch.cuda.synchronize()
    assert int(marks.sum()) == 0
    lb = b.loss_and_grads(x)
    ya, yb = a._signed.output(B).cpu().double(), b._signed.output(B).cpu().double()
    assert bool(torch.isfinite(ya).all()) and float((ya - yb).abs().max()) <= 1e-2 and float((ya - yb).abs().median()) <= 2e-4
    b2 = HipSquaredTrainer(plan_c, tensors, device=hip_device)  # (c's gradient alone, layer by layer)
    b2._signed.stage(x, stream)
    b2._signed.forward(B, stream)
    b2._flat_grad.zero_()
    b2._signed.backward(B, -2.0 / B, stream)
    torch.cuda.synchronize()
    for k in tensors:
        ga, gb = a.grads[k].double().cpu(), b2.grads[k].double().cpu()
        assert bool(torch.isfinite(ga).all()), k
        assert float((ga - gb).norm()) <= 2e-2 * float(gb.norm()) + 1e-12, (k, float((ga - gb).norm()), float(gb.norm()))


def _to_tile_native(x):
    """(..., B, 32) row-major -> (..., Bp * 32) in the tile-native order of ck_signed.hip (rows padded to whole 32-row tiles)."""
    *lead, B, K = x.shape
    Bp = (B + 31) // 32 * 32
    xp = torch.zeros(*lead, Bp, K, dtype=x.dtype, device=x.device)
    xp[..., :B, :] = x
    # dword (tile, g, kh, b, t) <- row 32 tile + b, unit 8 g + 4 kh + t
    return xp.reshape(*lead, Bp // 32, 32, 4, 2, 4).permute(*range(len(lead)), len(lead), len(lead) + 2, len(lead) + 3, len(lead) + 1,
                                                             len(lead) + 4).reshape(*lead, Bp * K).contiguous()


def _from_tile_native(y, B):
    *lead, n = y.shape
    Bp = n // 32
    x = y.reshape(*lead, Bp // 32, 4, 2, 32, 4).permute(*range(len(lead)), len(lead), len(lead) + 3, len(lead) + 1, len(lead) + 2,
                                                        len(lead) + 4).reshape(*lead, Bp, 32)
    return x[..., :B, :].contiguous()


@pytest.mark.gpu
def test_signed_log_layers_match_the_complex_layers(hip_device):
    """ck_slse_fwd / ck_slse_bwd against ck_sum_lse_fwd_c / ck_sum_lse_bwd_c on the same values (32 -> 32 and 32 -> 1 CP-T folds,
    ragged batch of 77 rows, negative weights and inputs): log|out| and sign = (re, im / pi) of the complex output, the children's
    gradient = the real part of the complex one, the same dW."""
    from cirkit_amd import _capi as capi

    g = torch.Generator().manual_seed(11)
    B, H = 77, 2
    Bp = (B + 31) // 32 * 32
    for F, Ko in ((3, 32), (2, 1)):
        w = (torch.randn(F, Ko, 32, generator=g) * 0.3).to(hip_device)
        mag = torch.randn(F * H, B, 32, generator=g).to(hip_device)
        neg = (torch.rand(F * H, B, 32, generator=g) < 0.4).to(hip_device)
        sw = ((neg.cpu().numpy().astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        signs_np = np.zeros((F * H, Bp), dtype=np.int32)  # (one word per row, blocks of Bp rows)
        signs_np[:, :B] = sw
        signs = torch.from_numpy(signs_np).to(hip_device)
        mag_n = _to_tile_native(mag)  # (F H, Bp 32): the layout the signed-log kernels read and write
        xc = torch.complex(mag, torch.pi * neg.to(torch.float32)).contiguous()
        ro_c = (torch.arange(F * H, dtype=torch.int64) * B * 32).reshape(F, H).to(hip_device)
        ro_n = (torch.arange(F * H, dtype=torch.int64) * Bp * 32).reshape(F, H).to(hip_device)
        stream = torch.cuda.current_stream().cuda_stream
        native_out = Ko == 32
        out_n = torch.zeros(F, Bp * 32 if native_out else B * Ko, device=hip_device)
        sout = torch.zeros(F, Bp if native_out else B, dtype=torch.int32, device=hip_device)
        capi.call("ck_slse_fwd", mag_n.data_ptr(), signs.data_ptr(), ro_n.data_ptr(), w.data_ptr(), out_n.data_ptr(), sout.data_ptr(), F, H, B, Ko,
                  None, None, None, None, None, 0, stream)
        outc = torch.zeros(F, B, Ko, dtype=torch.complex64, device=hip_device)
        capi.call("ck_sum_lse_fwd_c", xc.data_ptr(), ro_c.data_ptr(), w.data_ptr(), outc.data_ptr(), F, H, B, 32, Ko, capi.CK_SUM_PROD, 0, stream)
        torch.cuda.synchronize()
        out = _from_tile_native(out_n, B) if native_out else out_n.view(F, B, Ko)
        # random signs cancel: the yardstick is the sum of the magnitudes, sum_j |W_oj| |x_j|, in the linear domain
        bits = ((sout[:, :B].cpu().numpy().view(np.uint32)[..., None] >> np.arange(Ko, dtype=np.uint32)) & 1).astype(bool)
        v = mag.view(F, H, B, 32).sum(1).double()
        m = v.amax(-1, keepdim=True)
        yard = torch.einsum("foi,fbi->fbo", w.abs().double(), (v - m).exp())
        y_signed = torch.where(torch.from_numpy(bits).to(hip_device), -1.0, 1.0) * (out.double() - m).exp()
        y_complex = ((outc.to(torch.complex128) - m).exp()).real
        assert float(((y_signed - y_complex).abs() / yard).max()) <= 2e-6
        sure = ((y_complex.abs() / yard) > 1e-5).cpu().numpy()  # (the sign of a sum that cancelled to rounding noise is anybody's)
        assert np.array_equal(bits[sure], (np.abs(outc.imag.cpu().numpy()) > 1.5)[sure])
        gout = torch.randn(F, B, Ko, generator=g).to(hip_device)
        gout_n = _to_tile_native(gout) if native_out else gout.reshape(F, B * Ko).contiguous()
        gx_n, dw = torch.zeros(F, Bp * 32, device=hip_device), torch.zeros_like(w)  # (one gradient block per fold: its children share it)
        capi.call("ck_slse_bwd", mag_n.data_ptr(), signs.data_ptr(), ro_n.data_ptr(), w.data_ptr(), out_n.data_ptr(), sout.data_ptr(),
                  gout_n.data_ptr(), None, gx_n.data_ptr(), dw.data_ptr(), F, H, B, Ko, None, None, None, None, None, 0, stream)
        gxc, dwc = torch.zeros_like(xc), torch.zeros_like(w)
        goutc = torch.complex(gout, torch.zeros_like(gout)).contiguous()
        capi.call("ck_sum_lse_bwd_c", xc.data_ptr(), gxc.data_ptr(), ro_c.data_ptr(), w.data_ptr(), outc.data_ptr(), goutc.data_ptr(), dwc.data_ptr(),
                  F, H, B, 32, Ko, capi.CK_SUM_PROD, 0, stream)
        torch.cuda.synchronize()
        gx = _from_tile_native(gx_n, B)
        scale = float(gxc.real.abs().max())
        for h in range(H):
            assert float((gx - gxc.real.view(F, H, B, 32)[:, h]).abs().max()) <= 1e-4 * scale
        assert float((dw - dwc).abs().max()) <= 1e-4 * float(dwc.abs().max())


def test_tile_native_layout_helpers_round_trip():
    """(row r, unit u) sits at dword 1024 (r / 32) + 256 (u / 8) + 4 ((r % 32) + 32 ((u / 4) % 2)) + u % 4 (include/cirkit_hip.h)."""
    x = torch.arange(2 * 45 * 32, dtype=torch.float32).reshape(2, 45, 32)
    y = _to_tile_native(x)
    assert y.shape == (2, 64 * 32) and torch.equal(_from_tile_native(y, 45), x)
    for r, u in ((0, 0), (1, 0), (0, 5), (33, 12), (44, 31)):
        d = 1024 * (r // 32) + 256 * (u // 8) + 4 * ((r % 32) + 32 * ((u // 4) % 2)) + u % 4
        assert float(y[1, d]) == float(x[1, r, u]), (r, u)


@pytest.mark.gpu
def test_signed_squared_trainer_drops_a_batch_with_an_illegal_category(hip_device):
    """The signed-log path stages and checks the batch itself: an out-of-range category leaves parameters and moments alone."""
    from cirkit_amd.training_squared import HipSquaredTrainer

    plan_c = Plan.load(os.path.join(GOLDEN, "cfg5_sos_c_k32"))
    tensors = {k: np.where(v == 0, np.float32(1e-2), v).astype(np.float32) for k, v in init_plan_tensors(plan_c).items()}
    tr = HipSquaredTrainer(plan_c, tensors, device=hip_device, lr=1e-3)
    assert tr._signed is not None
    x = torch.randint(0, 256, (64, 784), generator=torch.Generator().manual_seed(3)).to(hip_device)
    for _ in range(4):
        tr.step(x)
    before = {k: v.copy() for k, v in tr.parameters().items()}
    bad = x.clone()
    bad[5, 100] = 256
    ll = tr.step(bad)
    assert not np.isfinite(float(ll[0]))
    after = tr.parameters()
    assert all(np.array_equal(before[k], after[k]) for k in before)
    assert tr.skipped_steps == 1 and tr.step_count == 4
    with pytest.raises(IndexError):
        tr.check_inputs()
    tr.step(x)
    now = tr.parameters()
    assert all(np.isfinite(v).all() for v in now.values()) and any(not np.array_equal(before[k], now[k]) for k in before)
    tr.check_inputs()


@pytest.mark.gpu
def test_real_squared_circuit_gradients_match_the_reference(hip_device):
    """A REAL circuit squared (Categorical inputs, CP-T, lse-sum): loss = -mean(2 c(x) - Z), Z built natively from the plan of c
    (ConstantValue layers over logs of Gram matrices of softmaxed rows, Hadamard, TensorDot) -- against the gradients of the
    reference's own `loss.backward()` (make_fixtures.py grads_sq_categorical)."""
    from cirkit_amd.training_squared import HipSquaredTrainer

    plan_c = Plan.load(os.path.join(GOLDEN, "sq_cat_qt4x4_k5"))
    with np.load(os.path.join(GOLDEN, "sq_cat_qt4x4_k5_grads.npz")) as z:
        ref = {k: z[k] for k in z.files}
    tensors = init_plan_tensors(plan_c, seed=6)
    tr = HipSquaredTrainer(plan_c, tensors, device=hip_device)
    x = torch.from_numpy(ref["x"].astype(np.int64)).to(hip_device)
    ll = tr.loss_and_grads(x).cpu().numpy()
    loss = -ll[0] / ll[1]
    assert abs(loss - float(ref["loss"])) <= 2e-5 * abs(float(ref["loss"])), (loss, float(ref["loss"]))
    got = tr.gradients()
    for k in tensors:
        want = ref["g_" + k]
        err = float(np.abs(got[k] - want).max())
        assert err <= 1e-3 * max(1e-3, float(np.abs(want).max())), (k, err, float(np.abs(want).max()))


@pytest.mark.gpu
def test_real_squared_gaussian_circuit_gradients_match_the_reference(hip_device):
    """Gaussian inputs, dense + CP-T layers under lse-sum, squared: the constant layers of the natively built Z hold closed-form
    log integrals of products of Gaussian units (`ck_param_gaussian_product_logz` and its backward) over pointer /
    scaled-sigmoid graphs -- against the reference's `loss.backward()` (make_fixtures.py grads_sq_gaussian)."""
    from cirkit_amd.training_squared import HipSquaredTrainer

    plan_c = Plan.load(os.path.join(GOLDEN, "sq_gauss_qt4x4_k4"))
    with np.load(os.path.join(GOLDEN, "sq_gauss_qt4x4_k4_grads.npz")) as z:
        ref = {k: z[k] for k in z.files}
    tensors = init_plan_tensors(plan_c, seed=8)
    tr = HipSquaredTrainer(plan_c, tensors, device=hip_device)
    x = torch.from_numpy(ref["x"].astype(np.float32)).to(hip_device)
    ll = tr.loss_and_grads(x).cpu().numpy()
    loss = -ll[0] / ll[1]
    assert abs(loss - float(ref["loss"])) <= 2e-5 * max(1.0, abs(float(ref["loss"]))), (loss, float(ref["loss"]))
    got = tr.gradients()
    for k in tensors:
        want = ref["g_" + k]
        err = float(np.abs(got[k] - want).max())
        assert err <= 1e-3 * max(1e-3, float(np.abs(want).max())), (k, err, float(np.abs(want).max()))


@pytest.mark.gpu
def test_squared_trainer_drops_a_batch_with_an_illegal_category(hip_device):
    """an out-of-range category (an IndexError in the reference) makes c(x) NaN; the step on that batch must
    change nothing (parameters, moments), `check_inputs()` reports it, and later batches train normally."""
    from cirkit_amd.training_squared import HipSquaredTrainer

    plan_c, _, tensors, ref = _case()
    tr = HipSquaredTrainer(plan_c, tensors, device=hip_device, lr=0.01)
    x = torch.from_numpy(ref["x"].astype(np.int64)).to(hip_device)
    tr.step(x)
    before = {k: v.copy() for k, v in tr.parameters().items()}
    bad = x.clone()
    bad[1, 2] = 10 ** 6
    tr.step(bad)
    after = tr.parameters()
    assert all(np.array_equal(before[k], after[k]) for k in before)
    assert tr.skipped_steps == 1
    with pytest.raises(IndexError):
        tr.check_inputs()
    tr.step(x)
    now = tr.parameters()
    assert all(np.isfinite(v).all() for v in now.values()) and any(not np.array_equal(before[k], now[k]) for k in before)
    tr.check_inputs()


@pytest.mark.gpu
def test_squared_trainer_zero_embedding_weight_nobody_selects(hip_device):
    """an Embedding weight that is exactly 0 and that no batch row selects has gradient 0 through c(x) (0 * d log w
    would be NaN); the flat gradient stays finite."""
    from cirkit_amd.training_squared import HipSquaredTrainer

    plan_c, _, tensors, ref = _case()
    x = torch.from_numpy(ref["x"].astype(np.int64))
    emb = next(l for l in plan_c.layers if l.type == "embedding")
    name = emb.params["weight"].nodes[0].config["tensor"]
    w = np.array(tensors[name], copy=True)  # (F, K, C)
    var0 = int(emb.scope_idx[0, 0])
    unused = next(c for c in range(w.shape[-1]) if c not in set(x[:, var0].tolist()))
    w[0, 0, unused] = 0.0
    tensors = dict(tensors)
    tensors[name] = w
    tr = HipSquaredTrainer(plan_c, tensors, device=hip_device)
    tr.c(x.to(hip_device))  # (the backward of c is what divides by w; Z's Gram matrices see the zero as a plain factor)
    tr._flat_grad.zero_()
    with torch.cuda.device(tr.device):
        tr._bwd_c.run(int(x.shape[0]), -2.0 / int(x.shape[0]), torch.cuda.current_stream(tr.device).cuda_stream)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(tr._flat_grad).all())


def _dp_squared_worker(rank, world, port, out_path, signed):
    import torch.distributed as dist

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cirkit_amd.distributed import init_from_env, shard_bounds
    from cirkit_amd.training_squared import HipSquaredTrainer

    init_from_env("gloo")  # both ranks share the one GPU of the test box; the exchange goes through gloo
    plan_c, tensors, x = _cfg5_case(64)
    lo, hi = shard_bounds(len(x), rank, world)
    tr = HipSquaredTrainer(plan_c, tensors, device="cuda:0", optimizer="sgd", lr=1e-4, signed=signed)
    for _ in range(3):
        tr.step(x[lo:hi].to("cuda:0"), global_batch=len(x))
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, **tr.parameters())
    dist.barrier()
    dist.destroy_process_group()


def _cfg5_case(B):
    plan_c = Plan.load(os.path.join(GOLDEN, "cfg5_sos_c_k32"))
    tensors = {k: np.where(v == 0, np.float32(1e-2), v).astype(np.float32) for k, v in init_plan_tensors(plan_c).items()}
    return plan_c, tensors, torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(9))


@pytest.mark.gpu
@pytest.mark.parametrize("signed", [False, True])
def test_squared_trainer_data_parallel_equals_single_process(hip_device, tmp_path, signed):
    """Two ranks, the batch sharded 32 + 32 (`global_batch`): each adds its rows' gradients of c and its share of Z's, ONE
    all-reduce of the flat gradient, the optimizer launch on every rank -- after three SGD steps the parameters are those of one
    process on the whole batch (which runs the optimizer inside its recorded list)."""
    import socket

    import torch.multiprocessing as mp

    from cirkit_amd.training_squared import HipSquaredTrainer

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp_sq.npz")
    mp.spawn(_dp_squared_worker, args=(2, port, out, signed), nprocs=2, join=True)
    plan_c, tensors, x = _cfg5_case(64)
    tr = HipSquaredTrainer(plan_c, tensors, device=hip_device, optimizer="sgd", lr=1e-4, signed=signed)
    for _ in range(3):
        tr.step(x.to(hip_device))
    torch.cuda.synchronize()
    one = tr.parameters()
    with np.load(out) as z:
        two = {k: z[k] for k in one}
    for k in one:
        moved = float(np.abs(one[k] - tensors[k]).max())
        assert moved > 0.0, k
        # (the two runs add the rows' gradients in different orders: fp32 rounding of the sums, a fraction of the step -- or, for
        #  a step of a few ulps, the last bit of the parameter itself: an ulp of an entry of magnitude 2 .. 4 is 2.4e-7)
        ulp = 1.2e-7 * max(1.0, float(np.abs(one[k]).max()))
        assert float(np.abs(one[k] - two[k]).max()) <= 2e-3 * moved + 2 * ulp, (k, float(np.abs(one[k] - two[k]).max()), moved)
