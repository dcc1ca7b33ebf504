"""HIP path vs the oracle and the committed reference outputs (GPU box only).

Tolerance: BASELINE.json's north_star asks for log-likelihoods within 1e-4 RELATIVE of the
reference in fp32; the checks below use REL = 1e-4 on the circuit output and a tighter absolute
bound per layer (fp32 round-off of exp/log at |LL| up to a few 1e3).
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case

pytestmark = pytest.mark.gpu
REL = 1e-4


def _x_of(plan, g):
    if "x" not in g:
        return None
    x = g["x"]
    return torch.from_numpy(x.astype(np.float32 if x.dtype.kind == "f" else np.int64))


def _check_layers(plan, tensors, x, hc, atol_scale=2e-6):
    from oracle.torch_oracle import as_torch, evaluate_plan

    y_ref, outs_ref = evaluate_plan(plan, as_torch(tensors), x, return_all=True)
    outs = hc.layer_outputs(None if x is None else x.to(hc.device))
    torch.cuda.synchronize()
    if plan.semiring == "complex-lse-sum":
        _check_complex_layers_locally(plan, tensors, x, hc, outs_ref, atol_scale)
        return y_ref
    for i, (a, b) in enumerate(zip(outs, outs_ref)):
        if a is None:  # fused away
            continue
        a = a.cpu()
        assert a.shape == b.shape, (i, plan.layers[i].type, a.shape, b.shape)
        scale = max(1.0, float(b.abs().max()))
        err = float((a - b).abs().max())
        assert err <= 16 * atol_scale * scale, (i, plan.layers[i].type, err, scale)
    return y_ref


def _check_complex_layers_locally(plan, tensors, x, hc, outs_ref, atol_scale):
    """complex-lse-sum: signed weights cancel inside the linear-space sums, so rounding differences
    of one layer are amplified by the next and two fp32 evaluations of the WHOLE circuit legitimately
    differ at ill-conditioned entries.  Each HIP layer is therefore checked in isolation through
    the reference's per-layer contract ``forward(x: (F,H,B,Ki)) -> (F,B,Ko)``: same fp32 inputs
    as the oracle's layer, ground truth = that layer in fp64 on those inputs, and the HIP result
    must be as accurate as the reference's own fp32 evaluation of the layer."""
    from oracle.torch_oracle import _CLSE, _layer_forward, _select, as_torch, eval_param

    tt = as_torch(tensors)
    t64 = {k: (v.to(torch.complex128) if v.is_complex() else v.double()) for k, v in tt.items()}
    for i, (spec, layer) in enumerate(zip(plan.layers, hc.layers)):
        if spec.inputs is None:
            xin = None
            if spec.type == "constant":
                got = layer.forward(1 if x is None else x.shape[0]).cpu()
            else:
                xi = x[..., torch.from_numpy(spec.scope_idx)].permute(1, 0, 2)
                got = layer.forward(xi.to(hc.device)).cpu()
            ref32 = outs_ref[i]
            with torch.no_grad():
                torch.set_default_dtype(torch.float64)
                try:
                    p64 = {pn: eval_param(pg, t64) for pn, pg in spec.params.items()}
                    arg = (1 if x is None else x.shape[0]) if spec.type == "constant" else xi
                    ref64 = _layer_forward(_CLSE, spec, p64, arg)
                finally:
                    torch.set_default_dtype(torch.float32)
        else:
            xin = _select(outs_ref, spec.inputs)  # (F, H, B, Ki) complex64, the oracle's own input
            got = layer.forward(xin.to(hc.device)).cpu()
            ref32 = outs_ref[i]
            with torch.no_grad():
                torch.set_default_dtype(torch.float64)
                try:
                    p64 = {pn: eval_param(pg, t64) for pn, pg in spec.params.items()}
                    ref64 = _layer_forward(_CLSE, spec, p64, xin.to(torch.complex128))
                finally:
                    torch.set_default_dtype(torch.float32)
        torch.cuda.synchronize()
        assert got.shape == ref32.shape, (i, spec.type)
        fin = torch.isfinite(ref64.real)
        assert torch.equal(torch.isfinite(got.real), fin), (i, spec.type)
        scale = max(1.0, float(ref64.real[fin].abs().max())) if fin.any() else 1.0
        d_mine = (got.real.double()[fin] - ref64.real[fin]).abs()
        d_ref = (ref32.real.double()[fin] - ref64.real[fin]).abs()
        assert float(d_mine.max()) <= 16 * atol_scale * scale + 4.0 * float(d_ref.max()), (
            i, spec.type, float(d_mine.max()), float(d_ref.max()))
        assert float(d_mine.mean()) <= 4 * atol_scale * scale + 4.0 * float(d_ref.mean()), (i, spec.type)
        # imaginary parts agree modulo 2*pi (branch of the complex log)
        ph = (torch.exp(1j * got.imag.double()[fin]) - torch.exp(1j * ref64.imag[fin])).abs()
        ph_ref = (torch.exp(1j * ref32.imag.double()[fin]) - torch.exp(1j * ref64.imag[fin])).abs()
        assert float(ph.max()) <= 1e-3 + 4.0 * float(ph_ref.max()), (i, spec.type, float(ph.max()), float(ph_ref.max()))


@pytest.mark.parametrize("name", ["cfg1_rbt8", "cfg2_qt784", "cfg2t_qt784_cpt16", "cfg4_pd784", "tucker_qt16_k6",
                                  "tucker4_qt16_k3", "quadgraph_6x6_k4", "rbt6_perfeature_k2", "pd_gauss_6x6_k4"])
@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("fuse", [False, True])
def test_real_configs_match_reference(hip_device, name, use_graph, fuse):
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case(name)
    x = _x_of(plan, g)
    hc = HipCircuit(plan, tensors, device=hip_device, use_graph=use_graph, fuse=fuse)
    if name == "cfg2_qt784" and fuse is not False:
        assert hc._groups and hc._groups[0].depth == (4 if fuse is True else fuse)
    y = hc(x.to(hip_device)).cpu()
    ref = torch.from_numpy(g["y_f32"])
    assert y.shape == ref.shape
    rel = float(((y - ref).abs() / ref.abs().clamp_min(1e-30)).max())
    assert rel <= REL, rel
    if "y_f64" in g:
        rel64 = float(((y.double() - torch.from_numpy(g["y_f64"])).abs() / torch.from_numpy(g["y_f64"]).abs()).max())
        assert rel64 <= REL, rel64
    # a second call replays the recorded program / graph
    y2 = hc(x.to(hip_device)).cpu()
    assert torch.equal(y, y2)
    _check_layers(plan, tensors, x, hc)


@pytest.mark.parametrize("name", ["quadtree_4x4_softmax0_k4", "quadtree_4x4_sigmoid_k4", "quadtree_4x4_softplus_k4", "quadtree_4x4_posclamp_k4"])
def test_sum_weight_activations_match_reference(hip_device, name):
    """Sum weights behind the other activations templates/utils.py:185-194 names -- softmax along the output units, sigmoid, softplus,
    positive-clamp (TorchSoftmaxParameter / TorchSigmoidParameter / TorchSoftplusParameter / TorchClampParameter, nodes.py:656-772):
    the reference's committed outputs and every layer against the oracle.  (Four units: padded to 32 where the activation has a
    zero pre-image -- not under the positive clamp, whose circuit runs on the shape-generic kernels.)"""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case(name)
    x = _x_of(plan, g)
    hc = HipCircuit(plan, tensors, device=hip_device)
    y = hc(x.to(hip_device)).cpu()
    for key in ("y_f32", "y_f64"):
        ref = torch.from_numpy(g[key]).double()
        assert float(((y.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max()) <= REL, key
    _check_layers(plan, tensors, x, hc)


@pytest.mark.parametrize("fuse", [1, 2, 3])
def test_leaf_fusion_capped_at_fewer_levels(hip_device, fuse):
    """`HipCircuit(fuse=n)`: the leaf region fused over n CP-T levels only (the levels above run layer by layer) -- BASELINE
    config 2 against the reference's outputs, every materialised layer against the oracle."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    x = _x_of(plan, g)
    hc = HipCircuit(plan, tensors, device=hip_device, fuse=fuse)
    assert hc._groups and hc._groups[0].depth == fuse
    y = hc(x.to(hip_device)).cpu()
    ref = torch.from_numpy(g["y_f32"])
    assert float(((y - ref).abs() / ref.abs().clamp_min(1e-30)).max()) <= REL
    _check_layers(plan, tensors, x, hc)


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLDEN, "kat_*.json"))))
def test_reference_known_answers(hip_device, name):
    """The reference's own KATs (tests/symbolic/test_utils.py:293-503 of the reference) under all
    four fold/optimize combinations: evidence values and the partition function by enumeration."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case(name)
    x = _x_of(plan, g)
    hc = HipCircuit(plan, tensors, device=hip_device, use_graph=False)
    y = hc(x.to(hip_device)).cpu().double().reshape(-1)
    ref = torch.from_numpy(g["y_f32"]).double().reshape(-1)
    assert float((y - ref).abs().max()) <= 1e-5
    for kx, ky in zip(g["kat_x"], g["kat_y"]):
        row = [i for i in range(len(x)) if np.allclose(x[i].numpy(), kx)]
        assert row, kx
        assert abs(float(torch.exp(y[row[0]])) - float(ky)) <= 2e-4 * max(1.0, abs(float(ky)))
    if "bernoulli" in name:  # sum over all 2^5 worlds == partition function 318.0
        assert abs(float(torch.exp(y).sum()) - float(g["kat_z"])) <= 1e-3 * float(g["kat_z"])
    _check_layers(plan, tensors, x, hc)


def test_sos_complex_circuit_and_partition(hip_device):
    """Config 5: c(x) under complex-lse-sum and Z = integral of |c|^2 share weights through
    pointer parameters; log p(x) = 2 Re c(x) - Re Z."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.parameters import TensorStore

    plan_c, tensors, gc = load_case("cfg5_sos_c_k32")
    plan_z, _, gz = load_case("cfg5_sos_z_k32")
    store = TensorStore(hip_device)
    store.update(tensors)
    hc = HipCircuit(plan_c, store, device=hip_device, use_graph=False)
    hz = HipCircuit(plan_z, store, device=hip_device, use_graph=False)
    x = _x_of(plan_c, gc)
    y = hc(x.to(hip_device)).cpu()
    z = hz().cpu()
    yr, zr = torch.from_numpy(gc["y_c64"]), torch.from_numpy(gz["z_c64"])
    assert y.shape == yr.shape and z.shape == zr.shape
    assert float(((y.real - yr.real).abs() / yr.real.abs()).max()) <= REL
    assert float(((z.real - zr.real).abs() / zr.real.abs()).max()) <= REL
    assert float((torch.exp(1j * y.imag) - torch.exp(1j * yr.imag)).abs().max()) <= 5e-3
    lp = 2 * y.real - z.real
    lpr = 2 * yr.real - zr.real
    assert float(((lp - lpr).abs() / lpr.abs()).max()) <= REL
    _check_layers(plan_c, tensors, x, hc)
    _check_layers(plan_z, tensors, None, hz)


def test_bf16_split_variants_of_a_wide_circuit_against_fp64(hip_device, capsys):
    """`contraction="bf16x3"` / `"bf16x6"` on a circuit whose sum layers have 128 units (QuadTree 8x8, Categorical-256, CP-T):
    the layer-wise dense / CP-T launches on bf16 pieces (`ck_sum_lse_fwd_v`), everything else exact.  Against the oracle's
    fp64 evaluation of the whole circuit."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    plan = image_data((1, 8, 8), "quad-tree-2", input_layer="categorical", num_input_units=128, sum_product_layer="cp-t",
                      num_sum_units=128)
    tensors = init_plan_tensors(plan, seed=4)
    x = torch.randint(0, 256, (200, 64), generator=torch.Generator().manual_seed(3))
    y64 = _fp64_outputs(plan, tensors, x)[0].reshape(-1).double()
    err = {}
    for c in ("f32", "bf16x3", "bf16x6"):
        hc = HipCircuit(plan, tensors, device=hip_device, contraction=c)
        y = hc(x.to(hip_device)).reshape(-1).double().cpu()
        err[c] = float(((y - y64).abs() / y64.abs()).max())
    with capsys.disabled():
        print(f"\n[bf16 variants, K = 128 circuit] max rel err vs fp64: " + ", ".join(f"{c} {e:.2e}" for c, e in err.items()))
    assert err["f32"] <= 1e-6 and err["bf16x6"] <= max(4.0 * err["f32"], 1e-6) and err["bf16x3"] <= 1e-4
    assert err["bf16x3"] > err["bf16x6"]  # (the variant launches are what ran)


@pytest.mark.parametrize("complex_sums", [False, True])
def test_complex_embedding_weights(hip_device, complex_sums):
    """Config 5's circuit c(x) with COMPLEX Embedding weights (the reference compiles DataType.COMPLEX tensors,
    rules/parameters.py:75-86; TorchEmbeddingLayer.forward maps them with torch.log of a complex number, input.py:258-266):
    the input layers gather a complex table and write (log|w|, arg w) (`ck_embedding_clog_c_fwd`), everything behind them
    runs on the complex kernels.  Against the oracle in complex128 with the oracle's own complex64 error as yardstick."""
    from cirkit_amd.circuit import HipCircuit
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, g = load_case("cfg5_sos_c_k32")
    rng = np.random.default_rng(5)
    emb = {n.config["tensor"] for sp in plan.layers if sp.type == "embedding" for n in sp.params["weight"].nodes if n.op == "tensor"}
    t2 = {}
    for k, v in tensors.items():
        v = np.asarray(v)
        if k in emb or (complex_sums and v.dtype.kind == "f"):
            v = (v * np.exp(1j * rng.uniform(-np.pi, np.pi, v.shape))).astype(np.complex64)  # (same moduli, random phases)
        t2[k] = v
    x = _x_of(plan, g)
    hc = HipCircuit(plan, t2, device=hip_device, use_graph=False)
    assert not hc._signed and not hc._emb_gather
    y = hc(x.to(hip_device)).cpu()
    y64 = _fp64_outputs(plan, t2, x)[0]
    y32 = evaluate_plan(plan, as_torch(t2), x)
    d_ref = float((y32.real.double() - y64.real).abs().max())
    assert torch.isfinite(y.real).all()
    assert float((y.real.double() - y64.real).abs().max()) <= 4.0 * d_ref + 1e-4 * float(y64.real.abs().max())
    assert float((torch.exp(1j * y.imag.double()) - torch.exp(1j * y64.imag)).abs().max()) <= 2e-2
    _check_layers(plan, t2, x, hc)


@pytest.mark.parametrize("name", ["sos_cat_c_qt4x4_k6", "sos_gauss_c_qt4x4_k4"])
def test_real_input_layers_in_a_complex_circuit(hip_device, name):
    """SURVEY.md 8d config 5 in its other form: Categorical (logits) -- and Gaussian -- input layers under
    complex-lse-sum (layers/input.py:276-278), signed sum weights.  Real part against the reference to 1e-4 of the
    value, the phase as a unit vector."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case(name)
    x = _x_of(plan, g)
    hc = HipCircuit(plan, tensors, device=hip_device, use_graph=False)
    y = hc(x.to(hip_device)).cpu()
    yr, y128 = torch.from_numpy(g["y_c64"]), torch.from_numpy(g["y_c128"])
    assert y.shape == yr.shape and y.dtype == torch.complex64
    # the reference's own fp32 run is this far from its fp64 run: the bar is REL or four times that
    own = float(((yr.real.double() - y128.real).abs() / y128.real.abs().clamp_min(1.0)).max())
    assert float(((y.real.double() - y128.real).abs() / y128.real.abs().clamp_min(1.0)).max()) <= max(REL, 4 * own)
    assert float((torch.exp(1j * y.imag.double()) - torch.exp(1j * y128.imag)).abs().max()) <= 5e-3
    _check_layers(plan, tensors, x, hc)


def _fp64_outputs(plan, tensors, x):
    """Every layer of the plan evaluated by the oracle in fp64 / complex128 (the whole circuit, un-isolated)."""
    from oracle.torch_oracle import as_torch, evaluate_plan

    t64 = {k: (v.to(torch.complex128) if v.is_complex() else v.double()) for k, v in as_torch(tensors).items()}
    torch.set_default_dtype(torch.float64)
    try:
        return evaluate_plan(plan, t64, x, return_all=True)
    finally:
        torch.set_default_dtype(torch.float32)


def test_real_valued_complex_circuit_whole_chain_against_fp64(hip_device):
    """Config 5 with the WHOLE chain under one check: the circuit's parameters are real, so c(x) is a real number and
    the fused launches work on signed linear tiles (ck_leaf.hip / ck_tail16.hip, signed).  Every materialised layer
    output -- not each layer in isolation -- against the oracle in fp64: Re within 1e-4 of the layer's scale, the phase
    as a unit vector, and never worse than twice the reference's own fp32 run."""
    from cirkit_amd.circuit import HipCircuit
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, g = load_case("cfg5_sos_c_k32")
    x = _x_of(plan, g)
    hc = HipCircuit(plan, tensors, device=hip_device)
    assert hc._signed and hc._groups and hc._tail
    y = hc(x.to(hip_device)).cpu()
    outs = hc.layer_outputs(x.to(hip_device))
    torch.cuda.synchronize()
    y64, outs64 = _fp64_outputs(plan, tensors, x)
    _, outs32 = evaluate_plan(plan, as_torch(tensors), x, return_all=True)
    def lin(z, m):  # the value relative to the largest magnitude of its row: what the next sum layer consumes
        return torch.exp(z.real - m) * torch.cos(z.imag)

    checked = 0
    for i, (a, r64, r32) in enumerate(zip(outs, outs64, outs32)):
        if a is None:
            continue
        a = a.cpu().to(torch.complex128)
        assert bool(((a.imag == 0) | ((a.imag - np.pi).abs() < 1e-6)).all())  # phases 0 / pi exactly on this path
        m = r64.real.amax(dim=-1, keepdim=True)
        z64, zh, z32 = lin(r64, m), lin(a, m), lin(r32.to(torch.complex128), m)
        err, own = float((zh - z64).abs().max()), float((z32 - z64).abs().max())
        # signed sums cancel: an entry is only as accurate as fp32 allows RELATIVE TO ITS ROW -- the measure is the
        # linear value over the row's largest magnitude, and the reference's own fp32 run is the yardstick
        assert err <= max(2e-5, 2 * own), (i, plan.layers[i].type, err, own)
        big = z64.abs() >= 0.1  # well-conditioned entries: also in log space, 1e-4 of the layer's scale
        scale = max(1.0, float(r64.real.abs().max()))
        assert float((a.real - r64.real)[big].abs().max()) <= 1e-4 * scale, (i, plan.layers[i].type)
        assert bool((torch.cos(a.imag)[big] * torch.cos(r64.imag)[big] > 0).all()), (i, "sign")
        checked += 1
    assert checked >= 2  # the roots of the fused leaf launch and the circuit output
    assert float(((y.real.double() - y64.real).abs() / y64.real.abs()).max()) <= REL


def test_signed_tiles_fall_back_to_log_space(hip_device):
    """Embedding rows that are one-hot at DIFFERENT units for sibling variables: every product of the first levels is
    tiny (1e-30 squared), far below the linear-space floor, yet a legitimate value in log space.  The signed leaf launch
    notes those tiles and evaluates them again in log space with signs: same result as the layer-by-layer complex kernels
    and the oracle."""
    from cirkit_amd.circuit import HipCircuit
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, g = load_case("cfg5_sos_c_k32")
    tensors = {k: np.array(v, copy=True) for k, v in tensors.items()}
    emb = next(l for l in plan.layers if l.type == "embedding")
    wname = emb.params["weight"].nodes[0].config["tensor"]
    w = tensors[wname]  # (784, 32, 256)
    F = w.shape[0]
    hot = np.full_like(w, 1e-30)
    sign = np.where(np.arange(F) % 3 == 0, -1.0, 1.0).astype(w.dtype)
    hot[np.arange(F), np.arange(F) % 32, :] = 1.0
    tensors[wname] = hot * sign[:, None, None]
    x = _x_of(plan, g)
    want = evaluate_plan(plan, as_torch(tensors), x)
    assert bool(torch.isfinite(want.real).all())
    hs = HipCircuit(plan, tensors, device=hip_device)
    hl = HipCircuit(plan, tensors, device=hip_device, signed_real=False)
    assert hs._signed and not hl._signed
    for _ in range(2):  # (twice: the marked-tile workspace must be clean again)
        ys, yl = hs(x.to(hip_device)).cpu(), hl(x.to(hip_device)).cpu()
        for got in (ys, yl):
            assert float(((got.real - want.real).abs() / want.real.abs().clamp_min(1.0)).max()) <= 1e-4
            assert float((torch.exp(1j * got.imag) - torch.exp(1j * want.imag)).abs().max()) <= 5e-3


@pytest.mark.parametrize("B", [1, 33, 100])
def test_signed_path_ragged_batches(hip_device, B):
    """The signed launches at batch sizes that are not multiples of their tiles, against the oracle and the
    layer-by-layer complex kernels."""
    from cirkit_amd.circuit import HipCircuit
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, _ = load_case("cfg5_sos_c_k32")
    gen = torch.Generator().manual_seed(B)
    x = torch.randint(0, 256, (B, 784), generator=gen)
    hs = HipCircuit(plan, tensors, device=hip_device)
    hl = HipCircuit(plan, tensors, device=hip_device, signed_real=False)
    want = evaluate_plan(plan, as_torch(tensors), x)
    ys, yl = hs(x.to(hip_device)).cpu(), hl(x.to(hip_device)).cpu()
    for got in (ys, yl):
        assert float(((got.real - want.real).abs() / want.real.abs()).max()) <= REL
        assert float((torch.exp(1j * got.imag) - torch.exp(1j * want.imag)).abs().max()) <= 5e-3


def test_mfma_and_generic_sum_kernels_agree(hip_device):
    from cirkit_amd import _capi as capi
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    x = _x_of(plan, g).to(hip_device)
    hc = HipCircuit(plan, tensors, device=hip_device, use_graph=False, fuse=False)
    y_fast = hc(x).clone()
    capi.call("ck_debug_force_generic", 1)
    try:
        y_gen = hc(x).clone()
    finally:
        capi.call("ck_debug_force_generic", 0)
    torch.cuda.synchronize()
    assert float(((y_fast - y_gen).abs() / y_gen.abs()).max()) <= 1e-6


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("B", [1, 31, 33, 100, 257])
def test_ragged_batch_sizes(hip_device, B, fuse):
    """Batch sizes that are not multiples of the 32-row MFMA tile / 256-row input tile."""
    from cirkit_amd.circuit import HipCircuit
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, tensors, g = load_case("cfg2t_qt784_cpt16")
    gen = torch.Generator().manual_seed(100 + B)
    x = torch.randint(0, 256, (B, 784), generator=gen)
    hc = HipCircuit(plan, tensors, device=hip_device, use_graph=False, fuse=fuse)
    y = hc(x.to(hip_device)).cpu()
    ref = evaluate_plan(plan, as_torch(tensors), x)
    assert float(((y - ref).abs() / ref.abs()).max()) <= REL
    plan2, tensors2, _ = load_case("cfg2_qt784")
    hc2 = HipCircuit(plan2, tensors2, device=hip_device, use_graph=False, fuse=fuse)
    y2 = hc2(x.to(hip_device)).cpu()
    ref2 = evaluate_plan(plan2, as_torch(tensors2), x)
    assert float(((y2 - ref2).abs() / ref2.abs()).max()) <= REL


@pytest.mark.parametrize("name", ["cfg2_qt784", "cfg4_pd784"])
def test_batched_and_per_node_parameter_paths_agree(hip_device, name):
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case(name)
    x = _x_of(plan, g).to(hip_device)
    a = HipCircuit(plan, tensors, device=hip_device, use_graph=False, fuse=False, batch_params=True)
    b = HipCircuit(plan, tensors, device=hip_device, use_graph=False, fuse=False, batch_params=False)
    assert len(a._bind(x.shape[0]) and a._batch) > 0
    ya, yb = a(x).cpu(), b(x).cpu()
    assert float(((ya - yb).abs() / yb.abs()).max()) <= 2e-6
    ref = torch.from_numpy(g["y_f32"])
    assert float(((yb - ref).abs() / ref.abs()).max()) <= REL


def test_dense_on_table_is_bit_identical(hip_device):
    """Applying the dense layer to the (F, C, K) table instead of to every batch row is the same
    arithmetic on the same values: outputs must be bit-for-bit equal."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    x = _x_of(plan, g).to(hip_device)
    for contraction in ("f32",):
        a = HipCircuit(plan, tensors, device=hip_device, use_graph=False, contraction=contraction, dense_on_table=True,
                       linear_levels=False)
        b = HipCircuit(plan, tensors, device=hip_device, use_graph=False, contraction=contraction, dense_on_table=False,
                       linear_levels=False)
        ya, yb = a(x).clone(), b(x).clone()
        torch.cuda.synchronize()
        assert torch.equal(ya, yb), float((ya - yb).abs().max())
        oa, ob = a.layer_outputs(x), b.layer_outputs(x)
        torch.cuda.synchronize()
        assert torch.equal(oa[5], ob[5])


@pytest.mark.parametrize("contraction", ["f32"])
def test_linear_levels_agree_with_log_space_levels(hip_device, contraction):
    """Chaining the fused levels in linear space (value = linear tile x per-row scale) computes the same
    sums as log -> exp between the levels: both within the 1e-4 bar of the reference's fp64 output, and
    the linear chain -- which skips a log / exp round trip per level -- at least as close to it."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    x = _x_of(plan, g).to(hip_device)
    ref = torch.from_numpy(g["y_f64"])
    lin = HipCircuit(plan, tensors, device=hip_device, contraction=contraction, linear_levels=True)
    log = HipCircuit(plan, tensors, device=hip_device, contraction=contraction, linear_levels=False)
    ya, yb = lin(x).cpu().double(), log(x).cpu().double()
    assert lin.kernel_label(lin._groups[0].root).startswith(("subtree_linear_kernel", "leaf_persistent_kernel"))
    assert log.kernel_label(log._groups[0].root).startswith("subtree_cat_cpt_kernel")
    ea, eb = float(((ya - ref) / ref).abs().max()), float(((yb - ref) / ref).abs().max())
    assert ea <= REL and eb <= REL
    assert ea <= 2.0 * eb + 1e-7, (ea, eb)
    # marginalised variables take the integral row of the linear table as well
    mask = torch.zeros(x.shape, dtype=torch.bool)
    mask[:, ::3] = True
    ma, mb = lin(x, integrate_vars=mask).cpu(), log(x, integrate_vars=mask).cpu()
    assert torch.allclose(ma, mb, rtol=1e-5, atol=1e-4)


def test_linear_levels_survive_products_at_the_edge_of_fp32(hip_device):
    """Children whose large units do not overlap: every product of a row underflows fp32 in linear space,
    the step is redone in log space (ck_fused.hip)."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import InputSpec, quad_tree_plan
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan = quad_tree_plan((1, 4, 4), input_layer=InputSpec("categorical", 8), sum_product="cp", num_input_units=32, num_sum_units=32)
    tensors = init_plan_tensors(plan, seed=2)
    # mixture components 0..15 put their mass on category 0, components 16..31 on category 1 (logit gap 80),
    # and the dense layer keeps the two halves apart (gap 80): for sibling pixels (0, 1) every unit of one
    # child is ~1 where the other is ~e^-80, so all 32 products are ~2e-35 -- below what the linear chain
    # accepts -- while the log-space value is perfectly finite
    half = (np.arange(32) < 16)
    t0 = np.full(tensors["t0"].shape, -80.0, dtype=np.float32)
    t0[:, half, 0] = 0.0
    t0[:, ~half, 1] = 0.0
    t1 = np.where(half[:, None] == half[None, :], 0.0, -80.0).astype(np.float32)
    tensors = {**tensors, "t0": t0, "t1": np.broadcast_to(t1, tensors["t1"].shape).copy()}
    x = torch.randint(0, 3, (64, 16), generator=torch.Generator().manual_seed(1))
    hc = HipCircuit(plan, tensors, device=hip_device)
    y = hc(x.to(hip_device)).cpu()
    assert hc._groups and hc._table_fused and hc.kernel_label(hc._groups[0].root, 64).startswith("subtree_linear_kernel")
    ref = evaluate_plan(plan, as_torch(tensors), x)
    assert torch.isfinite(ref).all() and float(ref.min()) < -150.0
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-3), float((y - ref).abs().max())


@pytest.mark.parametrize("B", [1, 33, 1000, 4096])
def test_persistent_leaf_launch_is_bit_identical(hip_device, B):
    """The fused leaf region as ONE resident workgroup per CU (ck_leaf.hip: weights staged once per segment, leaf rows
    gathered global -> LDS, next tile's inputs requested while the current one computes) computes exactly what the
    workgroup-per-128-rows launch computes: same arithmetic in the same order, bit for bit -- also with marginalised
    variables and a batch that is not a multiple of the 32-row tile."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    gen = torch.Generator().manual_seed(B)
    x = torch.randint(0, 256, (B, 784), generator=gen)
    x[::3, ::5] = -1
    x = x.to(hip_device)
    a = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=False)
    b = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=True)
    c = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=True, fuse=2)
    d = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=False, fuse=2)
    ya, yb = a(x).clone(), b(x).clone()
    assert a.kernel_label(a._groups[0].root, B).startswith("subtree_linear_kernel")
    assert b.kernel_label(b._groups[0].root, B).startswith("leaf_persistent_kernel")
    assert torch.equal(ya, yb)
    assert torch.equal(c(x), d(x))
    ref = HipCircuit(plan, tensors, device=hip_device, fuse=False)(x)
    assert torch.allclose(yb, ref, rtol=1e-5, atol=2e-3)


def test_persistent_leaf_segments_longer_than_a_chunk(hip_device, monkeypatch):
    """A segment of more than 64 x 8 tiles is walked in chunks (the 64-bit mask of tiles to redo covers one chunk): with
    whole roots as segments and 17 000 rows (532 tiles per segment) the launch is still bit-identical to the
    workgroup-per-128-rows launch."""
    import cirkit_amd.circuit_launch as circuit_mod  # (where `leaf_segments` is looked up by the leaf launches)
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    B = 17_000
    gen = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (B, 784), generator=gen)
    x[::11, ::3] = -1
    x = x.to(hip_device)
    monkeypatch.setattr(circuit_mod, "leaf_segments",
                        lambda num_roots, num_tiles, num_wg: np.asarray([[r, 0, num_tiles, 0] for r in range(num_roots)], dtype=np.int32))
    a = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=False)
    b = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=True)
    ya, yb = a(x).clone(), b(x).clone()
    assert b.kernel_label(b._groups[0].root, B).startswith("leaf_persistent_kernel")
    assert torch.equal(ya, yb)


def test_persistent_leaf_falls_back_to_log_space(hip_device):
    """Products at the edge of fp32 (see test_linear_levels_survive_products_at_the_edge_of_fp32) under the persistent
    launch: the tile is redone in log space by the same out-of-line walk, bit-identical to the other launch."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import InputSpec, quad_tree_plan
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan = quad_tree_plan((1, 8, 8), input_layer=InputSpec("categorical", 8), sum_product="cp", num_input_units=32, num_sum_units=32)
    tensors = init_plan_tensors(plan, seed=2)
    half = (np.arange(32) < 16)
    t0 = np.full(tensors["t0"].shape, -80.0, dtype=np.float32)
    t0[:, half, 0] = 0.0
    t0[:, ~half, 1] = 0.0
    t1 = np.where(half[:, None] == half[None, :], 0.0, -80.0).astype(np.float32)
    tensors = {**tensors, "t0": t0, "t1": np.broadcast_to(t1, tensors["t1"].shape).copy()}
    x = torch.randint(0, 3, (256, 64), generator=torch.Generator().manual_seed(1))
    a = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=False)
    b = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=True)
    ya, yb = a(x.to(hip_device)).cpu(), b(x.to(hip_device)).cpu()
    assert b.kernel_label(b._groups[0].root, 256).startswith("leaf_persistent_kernel")
    assert torch.equal(ya, yb)
    ref = evaluate_plan(plan, as_torch(tensors), x)
    assert torch.isfinite(ref).all()
    assert torch.allclose(yb, ref, rtol=1e-4, atol=1e-3), float((yb - ref).abs().max())


@pytest.mark.parametrize("B", [1, 17, 300, 4096])
def test_tail_on_16_row_tiles(hip_device, B):
    """The fused tail on 16-row tiles (ck_tail16.hip: descriptors and fold outputs in LDS, v_mfma_f32_16x16x4_f32)
    against the layer-wise evaluation (32-row tiles, one launch per layer): every tail layer's output agrees to fp32
    rounding (the two MFMA shapes add the 32 products in different orders; the leaf region below is evaluated in linear
    space by the fused circuit), and the log-likelihood sum folded into the launch equals the sum of its own outputs (fp64,
    deterministic)."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(B)).to(hip_device)
    a = HipCircuit(plan, tensors, device=hip_device, fuse=False)
    b = HipCircuit(plan, tensors, device=hip_device)
    assert b._tail16_ok() and b._tail_fuses_ll() and not a._tail
    la, lb = a.layer_outputs(x), b.layer_outputs(x)
    for j in b._tail:
        assert torch.allclose(la[j], lb[j], rtol=2e-6, atol=2e-6 * float(la[j].abs().max())), j
    y = b(x).clone()
    s1 = b.log_likelihood_sum(x).clone().cpu()
    s2 = b.log_likelihood_sum(x).clone().cpu()
    assert torch.equal(s1, s2)  # deterministic reduction
    assert s1[1].item() == B
    assert abs(s1[0].item() - float(y.double().sum())) <= 1e-9 * abs(float(y.double().sum()))
    sa = a.log_likelihood_sum(x).cpu()
    assert abs(sa[0].item() - s1[0].item()) <= 1e-6 * abs(s1[0].item())


@pytest.mark.parametrize("B", [1, 33, 1000, 4096])
def test_leaf_launch_reads_the_raw_batch(hip_device, B):
    """`direct_input` (default): when persistent leaf launches are the only readers of the discrete batch they read the
    caller's (B, D) int64 tensor themselves and no staged copy is made (`ck_leaf_walk_fwd` with a program input; the
    batch pointer of each call reaches the recorded launches through `ck_program_set_input`).  Same table rows, same
    arithmetic: bit-identical to the staged path -- ragged batches, marginalised entries, a different tensor per call,
    `forward` and `log_likelihood_sum`."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    a = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=True, direct_input=False)
    b = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=True)
    assert b.reads_batch_directly(B) and not a.reads_batch_directly(B)
    assert b.num_launches_ll(B) == a.num_launches_ll(B) - 1  # no staging launch
    for seed in range(3):  # three different tensors through the same recorded program
        x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(100 * B + seed))
        x[::3, ::5] = -1
        x[::7, 1::9] = -(2**31)  # the most negative sentinel the low dword still carries
        x = x.to(hip_device)
        ya, yb = a(x).clone(), b(x).clone()
        assert torch.equal(ya, yb)
        assert torch.equal(a.log_likelihood_sum(x), b.log_likelihood_sum(x))
        del x  # (the circuit keeps the batch of its last call alive)
    b.check_inputs()
    assert not b.reads_batch_directly(B) or b._bindings[B].x_last is not None


def test_forward_only_mode_keeps_results(hip_device):
    """`keep_layer_outputs=False` / `log_likelihood_sum`: the 32-unit folds of the tail that nobody outside it reads are
    not stored (ck_tail16_fold.skip_store: 24.7 MB per 4096-row batch at the north-star configuration) -- circuit outputs
    and the log-likelihood sum are unchanged, and `layer_outputs()` of the default circuit still sees every tail layer."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    B = 1000
    x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(9)).to(hip_device)
    a = HipCircuit(plan, tensors, device=hip_device)
    b = HipCircuit(plan, tensors, device=hip_device, keep_layer_outputs=False)
    ya = a(x).clone()
    assert torch.equal(ya, b(x))
    sa = a.log_likelihood_sum(x).clone()
    assert torch.equal(sa, b.log_likelihood_sum(x))
    la = a.layer_outputs(x)
    ref = HipCircuit(plan, tensors, device=hip_device, fuse=False).layer_outputs(x)
    for j in a._tail:  # (written by the forward program although log_likelihood_sum ran in between)
        assert torch.allclose(la[j], ref[j], rtol=1e-5, atol=2e-3), j


def test_bf16_split_contraction_variants_against_the_fp64_golden(hip_device, capsys):
    """`contraction="bf16x3"` / `"bf16x6"`: the depth-4 leaf launch with every fp32 operand of a contraction cut into two / three
    bf16 pieces and contracted on the bf16 matrix pipe (fp32 accumulation) -- labelled variants, never `value`.  Measured against
    the reference's float64 outputs at BASELINE config 2 (tests/golden/cfg2_qt784_golden.npz, y_f64), next to the exact-fp32
    path and to the reference's own fp32 run: bf16x6 must be fp32-like, bf16x3 within the 1e-4 bar; tiles that leave the linear
    range (read by the log-space walk from the fp32 weights) must still agree."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    x = torch.from_numpy(g["x"].astype(np.int64))
    y64 = torch.from_numpy(g["y_f64"]).reshape(-1).double()
    ref32 = float(((torch.from_numpy(g["y_f32"]).reshape(-1).double() - y64).abs() / y64.abs()).max())
    B = 4096  # (the persistent launch; the golden rows first)
    xb = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(5))
    xb[: x.shape[0]] = x
    xb = xb.to(hip_device)
    err, outs = {}, {}
    for c in ("f32", "bf16x3", "bf16x6"):
        hc = HipCircuit(plan, tensors, device=hip_device, contraction=c, persistent_leaf=True)
        y = hc(xb).reshape(-1).double().cpu()
        assert hc.kernel_label(hc._groups[0].root, B).startswith("leaf_persistent_kernel")
        outs[c] = y
        err[c] = float(((y[: x.shape[0]] - y64).abs() / y64.abs()).max())
    with capsys.disabled():
        print(f"\n[bf16 variants] max rel err vs the reference's fp64 output, config 2: reference fp32 {ref32:.2e}, "
              f"HIP f32 {err['f32']:.2e}, bf16x3 {err['bf16x3']:.2e}, bf16x6 {err['bf16x6']:.2e}")
    assert err["f32"] <= 1e-6 and err["bf16x6"] <= max(4.0 * err["f32"], 1e-6) and err["bf16x3"] <= 1e-4
    rel_all = float(((outs["bf16x6"] - outs["f32"]).abs() / outs["f32"].abs()).max())
    assert rel_all <= 2e-6  # (all 4096 rows against the exact path)
    with pytest.raises(ValueError):  # a variant of THAT launch only
        HipCircuit(plan, tensors, device=hip_device, contraction="bf16x3", persistent_leaf=True, direct_input=False)(xb)


@pytest.mark.parametrize("K", [32, 64])
def test_bf16_split_variants_of_a_tucker_circuit_against_fp64(hip_device, K, capsys):
    """`contraction="bf16x3"` / `"bf16x6"` on a circuit of Tucker layers (QuadGraph 8x8, Categorical-256, K units, batch 128: the
    notebook configuration in small): the stream-K Tucker launch contracts on the bf16 matrix instructions (`ck_tucker_fwd`),
    with the weights normalised online or by the prologue.  Against the oracle's fp64 evaluation of the whole circuit."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    plan = image_data((1, 8, 8), "quad-graph", input_layer="categorical", num_input_units=K, sum_product_layer="tucker",
                      num_sum_units=K)
    tensors = init_plan_tensors(plan, seed=3)
    x = torch.randint(0, 256, (128, 64), generator=torch.Generator().manual_seed(2))
    y64 = _fp64_outputs(plan, tensors, x)[0].reshape(-1).double()
    err = {}
    for fused in (True, False):
        for c in ("f32", "bf16x3", "bf16x6"):
            hc = HipCircuit(plan, tensors, device=hip_device, contraction=c, fused_weight_softmax=fused)
            y = hc(x.to(hip_device)).reshape(-1).double().cpu()
            if c != "f32":  # the variant did run: the results differ from the exact launch's in the last bits
                assert any("tucker" in hc.kernel_label(j, 128) for j in range(len(hc.layers)))
            err[fused, c] = float(((y - y64).abs() / y64.abs()).max())
    with capsys.disabled():
        print(f"\n[bf16 variants, Tucker circuit K={K}] max rel err vs fp64: " + ", ".join(f"{'logits' if f else 'weights'} {c} {e:.2e}" for (f, c), e in err.items()))
    for fused in (True, False):
        assert err[fused, "f32"] <= 1e-6 and err[fused, "bf16x6"] <= max(4.0 * err[fused, "f32"], 1e-6) and err[fused, "bf16x3"] <= 1e-4
        assert err[fused, "bf16x3"] > err[fused, "bf16x6"]  # (the three-product form is measurably coarser: it is what ran)


def test_shared_storage_sees_writes_through_data(hip_device):
    """`params_at_end=False` -- what `to_hip()` / `HipPipelineContext.compile(TorchCircuit)` pass, because the parameter storage
    is then SHARED with arbitrary torch code: a write through `p.data` (no version counter moves, `TensorStore.state()` cannot
    see it) reaches the very next forward, as in the reference, which re-evaluates its parameter graphs inside every forward
    (parameters/parameter.py:180-188).  Inference tensors (no version counter at all) are accepted the same way."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    B = 256
    x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(3)).to(hip_device)
    shared = {k: torch.from_numpy(np.array(v, copy=True)).to(hip_device) for k, v in tensors.items()}
    hc = HipCircuit(plan, shared, device=hip_device, pad_units=False, params_at_end=False)
    assert all(hc.store[k].data_ptr() == shared[k].data_ptr() for k in shared)  # (shared, not copied)
    y1 = hc(x).clone()
    name = list(shared)[2]
    shared[name].data.mul_(0.5)  # behind every counter
    y2 = hc(x).clone()
    fresh = HipCircuit(plan, {k: v.clone() for k, v in shared.items()}, device=hip_device, pad_units=False)(x)
    assert not torch.equal(y1, y2) and torch.equal(y2, fresh)
    with torch.inference_mode():
        inf = {k: v.clone() for k, v in shared.items()}
    hi = HipCircuit(plan, inf, device=hip_device, pad_units=False)  # (default params_at_end: falls back to re-evaluating at the start)
    assert torch.equal(hi(x), fresh) and torch.equal(hi(x), fresh)


@pytest.mark.parametrize("B", [33, 1000, 4096, 5000])
def test_parameters_evaluated_at_the_end_of_a_forward(hip_device, B):
    """`params_at_end=True`: the launch that walks the tail of a forward also re-evaluates the parameter graphs, for the next
    forward (`ck_tail_params_fwd`: tail blocks on 8 waves with their fold tiles in reused LDS slots, table-job blocks, 32-wide
    softmax blocks).  Same device functions as the two launches it replaces: tables, log scales, every weight, every tail
    layer output, the circuit output and the fused log-likelihood sum are bit-identical; a forward after `store.set` sees
    the new values (they are evaluated at its start); two launches per forward."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    tensors = {k: np.array(v, copy=True) for k, v in tensors.items()}
    kw = dict(device=hip_device, persistent_leaf=True)
    a = HipCircuit(plan, tensors, params_at_end=False, **kw)
    b = HipCircuit(plan, tensors, **kw)  # (the default)
    assert b._bind(B).params_at_end and not a._bind(B).params_at_end
    assert b.num_launches_ll(B) == a.num_launches_ll(B) - 1
    assert b._tail_slots()[2] < sum(b.layers[j].num_folds for j in b._tail)  # (slots are reused)
    rng = np.random.default_rng(B)
    for step in range(5):
        x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(13 * B + step))
        x[::5, ::7] = -1
        x = x.to(hip_device)
        la, lb = a.layer_outputs(x), b.layer_outputs(x)
        for j in b._tail:
            assert torch.equal(la[j], lb[j]), (step, j)
        ya = a(x).clone()
        assert torch.equal(ya, b(x)), step
        assert torch.equal(a.log_likelihood_sum(x), b.log_likelihood_sum(x)), step
        ra, rb = a._groups[0].root, b._groups[0].root
        assert torch.equal(a._group_dev[ra][1], b._group_dev[rb][1]) and torch.equal(a._group_dev[ra][3], b._group_dev[rb][3])
        for j in list(b._tail) + list(b._groups[0].levels):
            assert torch.equal(a.layers[j]._w, b.layers[j]._w), (step, j)
        if step in (1, 2):  # new parameter values: the next forward must use them
            for name in list(tensors)[:4]:
                new = a.store[name].cpu().numpy() + 0.1 * rng.standard_normal(tuple(a.store[name].shape)).astype(np.float32)
                a.store.set(name, new)
                b.store.set(name, new)
        if step == 3:  # an in-place torch operation on the stored tensors (what an optimizer does): seen through their version counters
            for name in list(tensors)[:4]:
                a.store[name].mul_(0.9)
                b.store[name].mul_(0.9)


def test_raw_batch_is_validated_row_by_row(hip_device):
    """The leaf launches look at low dwords only; the tail launch checks the full 64-bit values of its 16 rows
    (`ck_tail16_walk_fwd`): a row holding a category >= num_categories (an IndexError in the reference, input.py:399-412),
    a value whose LOW DWORD alone would pass for a category (2^32 + 5), or a value below -2^31 comes out NaN and
    `check_inputs()` raises once; every other row of the batch equals the clean evaluation, later batches are unaffected."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    B = 200
    hc = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=True)
    assert hc.reads_batch_directly(B)
    x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(3))
    x[5, 7] = -1
    good = hc(x.to(hip_device)).clone().reshape(-1)
    hc.check_inputs()
    assert bool(torch.isfinite(good).all())
    bad = x.clone()
    bad[3, 5] = 256            # first value past the table
    bad[40, 783] = 2**32 + 5   # low dword 5
    bad[41, 0] = -(2**31) - 1  # low dword 0x7fffffff
    bad[199, 300] = 2**40
    y = hc(bad.to(hip_device)).clone().reshape(-1)
    rows = torch.zeros(B, dtype=torch.bool)
    rows[[3, 40, 41, 199]] = True
    assert torch.equal(torch.isnan(y).cpu(), rows)
    assert torch.equal(y[~rows.to(hip_device)], good[~rows.to(hip_device)])
    assert bool(torch.isnan(hc.log_likelihood_sum(bad.to(hip_device))[0]))
    assert torch.equal(hc(x.to(hip_device)).reshape(-1), good)  # not sticky: the next batch is evaluated normally
    with pytest.raises(IndexError):
        hc.check_inputs()
    hc.check_inputs()  # cleared
    loose = HipCircuit(plan, tensors, device=hip_device, persistent_leaf=True, validate_inputs=False)
    assert loose.reads_batch_directly(B)
    assert bool(torch.isfinite(loose(bad.to(hip_device))).all())  # (memory-safe: such values select the integral row)
    loose.check_inputs()


@pytest.mark.parametrize("case", ["cfg1_rbt8", "cfg2_qt784"])
def test_out_of_range_category_is_an_error_not_a_number(hip_device, case):
    """``TorchCategoricalLayer`` raises IndexError on a category >= num_categories (advanced indexing, input.py:399-412).
    With a staged batch the staging kernel flags it on the device: the outputs of that batch -- and of every batch until
    the flag is looked at -- are NaN, `check_inputs()` raises IndexError and clears the flag; valid batches are untouched,
    and the marginalisation sentinel (negative) is not an error."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case(case)
    x = _x_of(plan, g).to(hip_device)
    hc = HipCircuit(plan, tensors, device=hip_device, direct_input=False)
    good = hc(x).clone()
    hc.check_inputs()
    assert bool(torch.isfinite(good).all())
    C = max(l.num_categories for l in hc.layers if hasattr(l, "num_categories"))
    bad = x.clone()
    bad[3, 5] = C
    y = hc(bad).clone()
    assert bool(torch.isnan(y).all())
    assert bool(torch.isnan(hc.log_likelihood_sum(bad)[0]))
    assert bool(torch.isnan(hc(x)).all())  # sticky until checked
    with pytest.raises(IndexError):
        hc.check_inputs()
    hc.check_inputs()  # cleared
    assert torch.equal(hc(x), good)
    m = x.clone()
    m[:, 2] = -1  # marginalised: not an error
    assert bool(torch.isfinite(hc(m)).all())
    hc.check_inputs()
    loose = HipCircuit(plan, tensors, device=hip_device, validate_inputs=False)
    assert bool(torch.isfinite(loose(bad)).all())  # (clamped to the last category, as before)


def test_ll_sum(hip_device):
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg1_rbt8")
    x = _x_of(plan, g).to(hip_device)
    hc = HipCircuit(plan, tensors, device=hip_device)
    s = hc.log_likelihood_sum(x).cpu()
    assert s[1].item() == x.shape[0]
    assert abs(s[0].item() - float(g["y_f64"].sum())) <= 1e-4 * abs(float(g["y_f64"].sum()))


# ---------------------------------------------------------------------------------------------
# natively built plans (tests/test_templates.py pins their structure against the reference; here
# they are EVALUATED: mixing layers, collapsed sums with MatMul weights, Tucker blocks, factorised
# multi-channel inputs, per-feature input families)
# ---------------------------------------------------------------------------------------------
def _native_plan_names():
    from test_templates import PLAN_ONLY  # (the HCLT plans have no native recipe: test_chow_liu_circuits_match_oracle)

    return PLAN_ONLY


def _random_batch(plan, B, seed):
    g = torch.Generator().manual_seed(seed)
    cols = []
    for l in plan.layers:
        if l.inputs is not None or l.scope_idx is None:
            continue
        for v in l.scope_idx[:, 0]:
            if l.type == "gaussian":
                cols.append((int(v), torch.randn(B, generator=g)))
            else:
                n = int(l.config.get("num_categories", l.config.get("num_states", 2)))
                cols.append((int(v), torch.randint(0, n, (B,), generator=g).float()))
    x = torch.zeros(B, plan.num_variables)
    for v, c in cols:
        x[:, v] = c
    kinds = {l.type for l in plan.layers if l.inputs is None}
    return x if "gaussian" in kinds else x.long()


# Every fixture with the default launches; layer by layer (`fuse=False`) only the fixtures that differ in WHICH layer kernels
# they reach (one per region graph / sum-product layer / input arrangement) -- VERDICT r5 #7: 62 -> 45 cases
_LAYERWISE_TOO = {"plan_ff4_r3_cp", "plan_lt5_r2_rand3_cpt", "plan_pd_1x12x12_delta6-3_cp", "plan_poondomingos_1x20x20_tucker",
                  "plan_poondomingos_2x17x9_cp", "plan_quadgraph_1x4x4_cp_nc3", "plan_quadgraph_1x5x7_cpt", "plan_quadgraph_1x6x6_tucker",
                  "plan_quadgraph_1x8x8_cp_mixFalse", "plan_quadtree2_2x3x5_cpt", "plan_quadtree4_1x4x4_tucker", "plan_quadtree4_7x7_cpt",
                  "plan_randombinarytree_1x12x13_cp", "plan_rbt6_perfeature_cp"}


@pytest.mark.parametrize("name,fuse", [(n, True) for n in _native_plan_names()] + [(n, False) for n in _native_plan_names() if n in _LAYERWISE_TOO])
def test_native_template_plans_match_oracle(name, fuse, hip_device):
    from cirkit_amd import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from test_templates import _rebuild

    plan = _rebuild(name)
    tensors = init_plan_tensors(plan, seed=3)
    x = _random_batch(plan, 48, seed=11)
    hc = HipCircuit(plan, tensors, device=hip_device, fuse=fuse, use_graph=fuse)
    y_ref = _check_layers(plan, tensors, x, hc)
    y = hc(x.to(hip_device)).cpu()
    assert y.shape == y_ref.shape
    assert torch.allclose(y, y_ref, rtol=REL, atol=1e-5), float((y - y_ref).abs().max())


def test_native_poon_domingos_784_matches_golden(hip_device):
    """BASELINE config 4 built natively (no plan fixture involved) reproduces the reference's
    committed log-likelihoods."""
    from cirkit_amd import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    _, _, g = load_case("cfg4_pd784")
    plan = image_data((1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64,
                      sum_product_layer="cp", num_sum_units=64)
    hc = HipCircuit(plan, init_plan_tensors(plan), device=hip_device)
    y = hc(torch.from_numpy(g["x"].astype(np.float32)).to(hip_device)).cpu().numpy().reshape(-1)
    ref = g["y_f64"].reshape(-1)
    assert np.allclose(y, ref, rtol=REL), float(np.abs(y - ref).max())


def test_cached_parameters_are_refreshed_when_values_change(hip_device):
    """`cache_params=True` (serving): derived parameters are computed on the first forward and after
    every parameter update, never in between -- results equal the recompute-every-forward default."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    x = _x_of(plan, g).to(hip_device)
    ref = HipCircuit(plan, tensors, device=hip_device)
    hc = HipCircuit(plan, tensors, device=hip_device, cache_params=True)
    y0 = hc(x).clone()
    assert torch.equal(y0, ref(x))
    assert torch.equal(hc(x), y0) and torch.equal(hc(x[:7]), y0[:7])  # replays (and a new batch size) reuse the cache
    bumped = {k: (v + 0.25 * np.sin(np.arange(v.size, dtype=np.float32)).reshape(v.shape)) for k, v in tensors.items()}
    hc.store.update(bumped)
    ref.store.update(bumped)
    y1 = hc(x).clone()
    assert torch.equal(y1, ref(x)) and not torch.equal(y1, y0)
    # in-place torch operations on a stored tensor are seen through its version counter (TensorStore.state) ...
    hc.store["t1"].mul_(0.5)
    ref.store["t1"].mul_(0.5)
    y2 = hc(x).clone()
    assert torch.equal(y2, ref(x)) and not torch.equal(y2, y1)
    # ... writes that bypass torch (a foreign kernel on the raw pointer) need an explicit invalidation: `.data` views share
    # the storage but not the version counter
    hc.store["t1"].data.mul_(2.0)
    ref.store["t1"].data.mul_(2.0)
    assert torch.equal(hc(x), y2)  # (not seen -- the documented contract, also of the default `params_at_end`)
    hc.invalidate_parameters()
    ref.store.touch()
    y3 = hc(x).clone()
    assert torch.equal(y3, ref(x)) and not torch.equal(y3, y2)
    # a circuit that evaluates its parameters at the start of every forward sees such writes without being told
    eager = HipCircuit(plan, tensors, device=hip_device, params_at_end=False)
    eager.store.update({k: hc.store.export(k) for k in tensors})
    assert torch.equal(eager(x), y3)
    eager.store["t1"].data.mul_(0.5)
    assert torch.equal(eager(x), y2)


def test_forwards_in_flight_on_two_streams(hip_device):
    """HipCircuitStreams: results of interleaved forwards on two streams equal the single-stream ones."""
    from cirkit_amd import HipCircuit, HipCircuitStreams

    plan, tensors, g = load_case("cfg2_qt784")
    x = _x_of(plan, g).to(hip_device)
    ref = HipCircuit(plan, tensors, device=hip_device)(x).clone()
    pool = HipCircuitStreams(plan, tensors, n=2, device=hip_device)
    outs = []
    for i in range(6):
        y, st = pool(x[: 64 - i])
        with torch.cuda.stream(st):
            outs.append(y.clone())
    pool.synchronize()
    for i, y in enumerate(outs):
        assert torch.equal(y, ref[: 64 - i])
    s, st = pool.log_likelihood_sum(x)
    pool.synchronize()
    assert abs(float(s[0]) - float(ref.double().sum())) <= 1e-6 * abs(float(ref.double().sum()))


@pytest.mark.parametrize("K,F,H,S,B", [(64, 3, 3, 2, 300), (64, 2, 1, 3, 33), (32, 5, 4, 2, 257), (32, 1, 2, 1, 4096)])
def test_region_kernels_agree_bit_for_bit(hip_device, K, F, H, S, B):
    """`ck_region_lse_fwd` through the C ABI on a synthetic arena: the launch that stages input tiles and weights by LDS
    DMA (region_dma_kernel) and the register-path launch (region_lse_kernel, taken under ck_debug_force_generic) do the
    same arithmetic in the same order.  Some slots carry no dense layer (plain slots)."""
    from cirkit_amd import _capi as capi

    g = torch.Generator().manual_seed(K + F + H + S)
    T = H * S
    arena = (torch.randn(F * T, B, K, generator=g) * 3 - 5).to(hip_device)
    row_off = (torch.arange(F * T, dtype=torch.int64) * (B * K)).reshape(F, T)
    row_off = row_off[:, torch.randperm(T, generator=g)].contiguous().to(hip_device)
    w = torch.softmax(torch.randn(F * T, K, K, generator=g), dim=-1).to(hip_device)
    addr = torch.tensor([w.data_ptr() + i * K * K * 4 for i in range(F * T)], dtype=torch.int64).reshape(F, T)
    if T > 1:
        addr[:, -1] = 0  # a plain slot
    addr = addr.to(hip_device)
    mw = torch.softmax(torch.randn(F, K, H, generator=g), dim=-1).to(hip_device)
    outs = []
    stream = torch.cuda.current_stream(hip_device).cuda_stream
    for force in (0, 1):
        out = torch.full((F, B, K), float("nan"), device=hip_device)
        capi.call("ck_debug_force_generic", force)
        try:
            capi.call("ck_region_lse_fwd", arena.data_ptr(), row_off.data_ptr(), addr.data_ptr(), mw.data_ptr(),
                      out.data_ptr(), None, None, None, 0, None, F, H, S, B, K, stream)
        finally:
            capi.call("ck_debug_force_generic", 0)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    # and the value: log sum_h mw[:, h] prod_s G_{h,s}
    a, ro, wc, ad = arena.cpu().double(), row_off.cpu(), w.cpu().double(), addr.cpu()
    ref = torch.zeros(F, B, K, dtype=torch.float64)
    for f in range(F):
        acc = torch.zeros(B, K, dtype=torch.float64)
        for h in range(H):
            p = torch.zeros(B, K, dtype=torch.float64)
            for s_ in range(S):
                t = h * S + s_
                v = a[int(ro[f, t]) // (B * K)]
                if int(ad[f, t]) != 0:
                    wi = (int(ad[f, t]) - w.data_ptr()) // (K * K * 4)
                    v = torch.log(torch.exp(v) @ wc[wi].T)
                p = p + v
            acc = acc + mw[f, :, h].cpu().double() * torch.exp(p)
        ref[f] = torch.log(acc)
    assert float(((outs[0].double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= 1e-5
    # linear-space products (redo workspace given): same values to fp32 rounding, no workgroup marked
    redo = torch.zeros(F * ((B + 127) // 128), dtype=torch.int32, device=hip_device)
    out = torch.full((F, B, K), float("nan"), device=hip_device)
    capi.call("ck_region_lse_fwd", arena.data_ptr(), row_off.data_ptr(), addr.data_ptr(), mw.data_ptr(),
              out.data_ptr(), None, None, None, 0, redo.data_ptr(), F, H, S, B, K, stream)
    torch.cuda.synchronize()
    assert int(redo.abs().sum()) == 0
    assert float(((out.cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= 1e-5
    # the labelled bf16-split variants of the DMA-staged launch (ck_region_lse_fwd_v), log-space and linear-space form: bf16x6 as
    # close to the fp64 value as the exact launch; bf16x3 drops terms <= 2^-15 of a product, S + 1 contractions deep: 5e-4 of a
    # layer output of size ~3 (the 1e-4 bar is that of a circuit's output, |log-likelihood| ~ 1e3: test_bf16_split_*)
    def rel(o):
        return float(((o.cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max())

    e0 = rel(outs[0])
    for ct, bar in ((3, 5e-4), (6, max(4.0 * e0, 1e-6))):
        for ws in (None, redo):
            out = torch.full((F, B, K), float("nan"), device=hip_device)
            capi.call("ck_region_lse_fwd_v", arena.data_ptr(), row_off.data_ptr(), addr.data_ptr(), mw.data_ptr(), out.data_ptr(),
                      None, None, None, 0, None if ws is None else ws.data_ptr(), F, H, S, B, K, ct, stream)
            torch.cuda.synchronize()
            assert int(redo.abs().sum()) == 0
            assert rel(out) <= bar, (ct, ws is None, rel(out), e0)
            if ct == 3:
                assert rel(out) > e0  # (the three-product form is measurably coarser: it is what ran)


@pytest.mark.parametrize("post,gather", [(False, False), (True, False), (True, True), (False, True)])
@pytest.mark.parametrize("K,F,S,B", [(64, 3, 2, 300), (64, 2, 3, 33), (32, 5, 2, 257), (32, 1, 1, 4096)])
def test_cp_block_launches_agree_bit_for_bit(hip_device, K, F, S, B, post, gather):
    """`ck_cp_lse_fwd` for a CP block of one child per slot with contiguous output, with and without a CP-T sum behind the
    product, with and without slots that gather the rows of a table by the batch values (tabulated dense layers): the
    launch on the DMA-staged region kernel and the register-path launch (cp_lse_kernel, taken under
    ck_debug_force_generic) do the same arithmetic in the same order; value = [W_post .] prod_s (W_s . x_s) in log space."""
    from cirkit_amd import _capi as capi

    g = torch.Generator().manual_seed(K + F + S + B)
    C, D = 7, 5
    arena = (torch.randn(F * S, B, K, generator=g) * 3 - 5).to(hip_device)
    row_off = (torch.arange(F * S, dtype=torch.int64) * (B * K)).reshape(F, S)
    row_off = row_off[:, torch.randperm(S, generator=g)].contiguous().to(hip_device)
    w = torch.softmax(torch.randn(F * S, K, K, generator=g), dim=-1).to(hip_device)
    addr = torch.tensor([w.data_ptr() + i * K * K * 4 for i in range(F * S)], dtype=torch.int64).reshape(F, S)
    if S > 2:
        addr[:, 1] = 0  # a plain slot
    # gather slots: slot 0 of every fold reads row x[b, var] of its own (C + 1, K) table (negative = the integral row C)
    tab = (torch.randn(F, C + 1, K, generator=g) * 2 - 3).to(hip_device)
    xt = torch.randint(-1, C, (D, B), generator=g, dtype=torch.int32).to(hip_device)
    g_var = torch.full((F, S), -1, dtype=torch.int32)
    g_addr = torch.zeros(F, S, dtype=torch.int64)
    if gather:
        g_var[:, 0] = torch.arange(F, dtype=torch.int32) % D
        g_addr[:, 0] = torch.tensor([tab.data_ptr() + f * (C + 1) * K * 4 for f in range(F)], dtype=torch.int64)
        addr[:, 0] = 0  # (tabulated: no weights of their own)
    addr, g_var, g_addr = addr.to(hip_device), g_var.to(hip_device), g_addr.to(hip_device)
    gargs = (g_addr.data_ptr(), g_var.data_ptr(), xt.data_ptr(), C) if gather else (None, None, None, 0)
    wp = torch.softmax(torch.randn(F, K, K, generator=g), dim=-1).to(hip_device)
    paddr = torch.tensor([wp.data_ptr() + i * K * K * 4 for i in range(F)], dtype=torch.int64).to(hip_device)
    stream = torch.cuda.current_stream(hip_device).cuda_stream
    outs = []
    for force in (0, 1):
        out = torch.full((F, B, K), float("nan"), device=hip_device)
        capi.call("ck_debug_force_generic", force)
        try:
            capi.call("ck_cp_lse_fwd", arena.data_ptr(), row_off.data_ptr(), addr.data_ptr(), paddr.data_ptr() if post else None, None, out.data_ptr(),
                      *gargs, F, S, 1, B, K, stream)
        finally:
            capi.call("ck_debug_force_generic", 0)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    a, ro, wc, ad = arena.cpu().double(), row_off.cpu(), w.cpu().double(), addr.cpu()
    tc, xc, gv = tab.cpu().double(), xt.cpu().long(), g_var.cpu()
    ref = torch.zeros(F, B, K, dtype=torch.float64)
    for f in range(F):
        for s_ in range(S):
            if int(gv[f, s_]) >= 0:
                xb = xc[int(gv[f, s_])]
                v = tc[f][torch.where(xb < 0, torch.full_like(xb, C), xb)]
            else:
                v = a[int(ro[f, s_]) // (B * K)]
            if int(ad[f, s_]) != 0:
                wi = (int(ad[f, s_]) - w.data_ptr()) // (K * K * 4)
                v = torch.log(torch.exp(v) @ wc[wi].T)
            ref[f] += v
        if post:
            mx = ref[f].amax(dim=-1, keepdim=True)
            ref[f] = torch.log(torch.exp(ref[f] - mx) @ wp[f].cpu().double().T) + mx
    assert float(((outs[0].double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= 1e-5
    # the labelled bf16-split variants (ck_cp_lse_fwd_v; blocks of at most 8 slots, or without gathers, take the DMA-staged launch)
    e0 = float(((outs[0].double() - ref).abs() / ref.abs().clamp_min(1.0)).max())
    for ct, bar in ((3, 5e-4), (6, max(4.0 * e0, 1e-6))):
        out = torch.full((F, B, K), float("nan"), device=hip_device)
        capi.call("ck_cp_lse_fwd_v", arena.data_ptr(), row_off.data_ptr(), addr.data_ptr(), paddr.data_ptr() if post else None, None,
                  out.data_ptr(), *gargs, F, S, 1, B, K, ct, stream)
        torch.cuda.synchronize()
        err = float(((out.cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max())
        assert err <= bar, (ct, err, e0)


@pytest.mark.parametrize("K", [64, 32])
def test_region_linear_space_falls_back_to_log_space(hip_device, K):
    """Rows whose products leave the fp32 range in linear space (factors with disjoint supports, all -inf inputs, +inf):
    the linear-space launch marks their workgroups and the log-space launch evaluates them again -- same bits as the
    log-space evaluation alone; the workspace is clean afterwards (the program can be replayed)."""
    from cirkit_amd import _capi as capi

    F, H, S, B = 2, 2, 2, 700
    T = H * S
    g = torch.Generator().manual_seed(K)
    arena = torch.randn(F * T, B, K, generator=g) * 2 - 3
    # fold 0, rows 40..49: the two children of every partitioning are (almost) one-hot at different units and the
    # weights of fold 0 are identity matrices, so every product is exp(-60) or exp(-120): representable in log space,
    # below the linear-space floor
    for blk, unit in ((0, 3), (1, 7), (2, 11), (3, 13)):
        arena[blk, 40:50] = -60.0
        arena[blk, 40:50, unit] = 0.0
    arena[2, 300] = float("-inf")       # an impossible row
    arena[3, 301, 5] = float("inf")
    arena = arena.to(hip_device)
    row_off = (torch.arange(F * T, dtype=torch.int64) * (B * K)).reshape(F, T).to(hip_device)
    w = torch.softmax(torch.randn(F * T, K, K, generator=g) * 2, dim=-1)
    w[:T] = torch.eye(K)
    w = w.to(hip_device)
    addr = torch.tensor([w.data_ptr() + i * K * K * 4 for i in range(F * T)], dtype=torch.int64).reshape(F, T).to(hip_device)
    mw = torch.softmax(torch.randn(F, K, H, generator=g), dim=-1).to(hip_device)
    stream = torch.cuda.current_stream(hip_device).cuda_stream
    exact = torch.full((F, B, K), float("nan"), device=hip_device)
    capi.call("ck_region_lse_fwd", arena.data_ptr(), row_off.data_ptr(), addr.data_ptr(), mw.data_ptr(),
              exact.data_ptr(), None, None, None, 0, None, F, H, S, B, K, stream)
    redo = torch.zeros(F * ((B + 127) // 128), dtype=torch.int32, device=hip_device)
    for _ in range(2):  # twice: the workspace must be clean again
        out = torch.full((F, B, K), float("nan"), device=hip_device)
        capi.call("ck_region_lse_fwd", arena.data_ptr(), row_off.data_ptr(), addr.data_ptr(), mw.data_ptr(),
                  out.data_ptr(), None, None, None, 0, redo.data_ptr(), F, H, S, B, K, stream)
        torch.cuda.synchronize()
        assert int(redo.abs().sum()) == 0
        e, o = exact.cpu(), out.cpu()
        rows_per_wg = 128 if K == 64 else 256
        marked = {(0, 40 // rows_per_wg), (0, 300 // rows_per_wg), (0, 301 // rows_per_wg)}
        for f in range(F):
            for wg in range((B + rows_per_wg - 1) // rows_per_wg):
                a, b = e[f, wg * rows_per_wg:(wg + 1) * rows_per_wg], o[f, wg * rows_per_wg:(wg + 1) * rows_per_wg]
                if (f, wg) in marked:
                    assert torch.equal(a.isnan(), b.isnan()) and torch.equal(a[~a.isnan()], b[~b.isnan()]), (f, wg)
                else:
                    assert torch.allclose(a, b, rtol=1e-5, atol=1e-4), (f, wg)
    assert torch.isfinite(e[0, 40:50]).all() and float(e[0, 40:50].max()) < -50  # a legitimate, very small, value


@pytest.mark.parametrize("rg,shape,sp,K,inp,B", [
    ("poon-domingos", (1, 8, 8), "cp", 32, "gaussian", 37),      # region_lse_kernel<1>, cp blocks, gaussian products
    ("poon-domingos", (1, 8, 8), "cp", 64, "categorical", 130),  # region_lse_kernel<2>, leftovers, subsets
    ("poon-domingos", (2, 8, 8), "cp", 64, "gaussian", 33),      # two channels: factorised inputs
    ("quad-graph", (1, 8, 8), "cp", 32, "categorical", 70),
    ("quad-graph", (1, 7, 9), "cp", 64, "gaussian", 65),
    ("quad-graph", (1, 8, 8), "cp-t", 64, "categorical", 40),    # mixing layers without CP blocks
    ("quad-tree-2", (1, 8, 8), "cp", 64, "categorical", 96),     # K = 64 dense / CP-T on the tile kernels
    ("quad-tree-2", (1, 8, 8), "cp", 128, "categorical", 50),    # sum_lse_gemm_kernel
    ("quad-tree-4", (1, 8, 8), "cp-t", 96, "gaussian", 35),      # arity-4 CP-T, 96 units
    ("random-binary-tree", (1, 5, 5), "cp", 32, "categorical", 64),
])
@pytest.mark.parametrize("use_mixing", [True, False])
def test_template_circuits_on_the_mfma_kernels(hip_device, rg, shape, sp, K, inp, B, use_mixing):
    """Region-graph templates at unit counts that take the MFMA kernels (32, 64, 96, 128) -- the committed
    reference plans of these shapes use 2-4 units and only reach the shape-generic kernels -- against the
    oracle, layer by layer where the layer is materialised."""
    from cirkit_amd import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    if not use_mixing and rg in ("quad-tree-2", "quad-tree-4", "random-binary-tree"):
        pytest.skip("no regions with several partitionings")
    plan = image_data(shape, rg, input_layer=inp, num_input_units=K, sum_product_layer=sp, num_sum_units=K,
                      use_mixing_weights=use_mixing)
    tensors = init_plan_tensors(plan, seed=5)
    x = _random_batch(plan, B, seed=17)
    if inp == "categorical":
        x = torch.randint(0, 256, x.shape, generator=torch.Generator().manual_seed(3))
    hc = HipCircuit(plan, tensors, device=hip_device)
    y_ref = _check_layers(plan, tensors, x, hc)
    y = hc(x.to(hip_device)).cpu()
    assert torch.allclose(y, y_ref, rtol=REL, atol=1e-4), float((y - y_ref).abs().max())
    y2 = HipCircuit(plan, tensors, device=hip_device, fuse=False)(x.to(hip_device)).cpu()
    assert torch.allclose(y2, y_ref, rtol=REL, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("kw,K", [(dict(region_graph="quad-tree-2", sum_product_layer="cp"), 8),
                                  (dict(region_graph="quad-graph", sum_product_layer="cp"), 20),
                                  (dict(region_graph="quad-graph", sum_product_layer="tucker"), 12),
                                  (dict(region_graph="poon-domingos", sum_product_layer="cp", input_layer="gaussian"), 48),
                                  (dict(region_graph="quad-tree-2", sum_product_layer="cp", num_classes=10), 100)])
def test_padded_units_match_oracle(hip_device, kw, K):
    """Unit counts that are not multiples of 32 are padded to MFMA tiles (cirkit_amd/padding.py): same
    log-likelihoods as the oracle on the user's plan and as the unpadded HIP path; parameter updates and
    exports keep the user's shapes."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan = image_data((1, 8, 8), num_input_units=K, num_sum_units=K, **kw)
    tensors = init_plan_tensors(plan)
    g = torch.Generator().manual_seed(K)
    x = torch.randn((70, 64), generator=g) if kw.get("input_layer") == "gaussian" else torch.randint(0, 256, (70, 64), generator=g)
    want = evaluate_plan(plan, as_torch(tensors), x)
    hc = HipCircuit(plan, tensors, device=hip_device)
    assert hc._pad_info is not None and all(l.num_output_units in (1, 32, 64, 128) for l in hc.plan.layers)
    got = hc(x.to(hip_device)).cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= REL * float(want.abs().max())
    plain = HipCircuit(plan, tensors, device=hip_device, pad_units=False)
    assert plain._pad_info is None
    assert float((plain(x.to(hip_device)).cpu() - want).abs().max()) <= REL * float(want.abs().max())
    # a parameter update in the user's shape reaches the padded storage; export gives the user's shape back
    name = sorted(plan.tensors)[1]
    new = tensors[name] * 0.5 + 0.1
    hc.store.set(name, new)
    assert np.array_equal(hc.store.export(name), new.astype(np.float32))
    t2 = dict(tensors)
    t2[name] = new
    want2 = evaluate_plan(plan, as_torch(t2), x)
    got2 = hc(x.to(hip_device)).cpu()
    assert float((got2 - want2).abs().max()) <= REL * float(want2.abs().max())


def test_shared_store_of_a_padded_circuit(hip_device):
    """A second circuit over the parameter store of a padded one: the same plan is padded the same way; a plan that
    cannot be (here: its squared-partition plan, built from einsum / conj nodes) is refused with a clear message."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.functional import squared_partition_plan
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data

    plan = image_data((1, 4, 4), "quad-tree-2", num_input_units=6, num_sum_units=6)
    hc = HipCircuit(plan, init_plan_tensors(plan), device=hip_device)
    assert hc._pad_info is not None
    x = torch.randint(0, 256, (40, 16)).to(hip_device)
    y = hc(x).clone()
    plan_again = image_data((1, 4, 4), "quad-tree-2", num_input_units=6, num_sum_units=6)
    other = HipCircuit(plan_again, hc.store, device=hip_device)
    assert other._pad_info is not None and torch.equal(other(x), y)
    try:
        zplan = squared_partition_plan(plan)
    except (ValueError, NotImplementedError):
        return  # (squares of this layer mix are not built natively: nothing more to check)
    with pytest.raises(ValueError, match="pad_units=False"):
        HipCircuit(zplan, hc.store, device=hip_device)


@pytest.mark.parametrize("name", ["plan_clt_cat9_cp", "plan_clt_gauss7_cpt", "plan_clt_mixed6_cp"])
def test_chow_liu_circuits_match_oracle(hip_device, name):
    """HCLT circuits -- plans the REFERENCE compiled from structures it learned from the committed data
    (tests/golden/make_fixtures.py chow_liu; this backend has no structure learner): partitions of any arity, mixed input
    families -- on the HIP path against the oracle, evaluated on the rows the structures were learned from."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.plan import Plan
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan = Plan.load(os.path.join(GOLDEN, name))
    with np.load(os.path.join(GOLDEN, name + "_data.npz")) as z:
        x = torch.from_numpy(z["data"])
    tensors = init_plan_tensors(plan, seed=9)
    want = evaluate_plan(plan, as_torch(tensors), x)
    got = HipCircuit(plan, tensors, device=hip_device)(x.to(hip_device)).cpu()
    assert got.shape == want.shape and torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= REL * max(1.0, float(want.abs().max()))


def test_binomial_inputs_match_reference_on_the_gpu(hip_device):
    """Binomial input layers: reference outputs (incl. the ends of the support, x = 0 and x = 255), a marginal query,
    and a padded run."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from oracle.torch_oracle import as_torch, evaluate_plan

    plan, _, g = load_case("binomial_qg6x6_k4")
    tensors = init_plan_tensors(plan, seed=12)
    x = torch.from_numpy(g["x"].astype(np.int64))
    ref64 = torch.from_numpy(g["y_f64"])
    for pad in (False, True):
        hc = HipCircuit(plan, tensors, device=hip_device, pad_units=pad)
        y = hc(x.to(hip_device)).cpu()
        assert float(((y.double() - ref64).abs() / ref64.abs()).max()) <= REL
    mask = torch.zeros(36, dtype=torch.bool)
    mask[[3, 17, 30]] = True
    want = evaluate_plan(plan, as_torch(tensors), x, integrate_mask=mask.unsqueeze(0).expand(12, 36))
    got = hc(x.to(hip_device), integrate_vars=mask).cpu()
    assert float((got - want).abs().max()) <= REL * float(want.abs().max())


def test_padded_circuits_far_in_the_tails(hip_device):
    """Rows where every real input unit is astronomically unlikely (a Gaussian 40 sigma out, a Binomial at the ends of
    its support): the padded HIP circuit (input units padded with COPIES of real units) returns the oracle's finite
    value -- a dummy padded unit would win the row maximum and underflow everything real."""
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors
    from cirkit_amd.templates import image_data
    from oracle.torch_oracle import as_torch, evaluate_plan

    for kw, x in ((dict(input_layer="gaussian"), torch.full((33, 64), 40.0)),
                  (dict(input_layer="binomial"), torch.tensor([[0] * 64, [255] * 64, [128] * 64]))):
        plan = image_data((1, 8, 8), "quad-graph", num_input_units=5, num_sum_units=5, **kw)
        tensors = init_plan_tensors(plan, seed=2)
        want = evaluate_plan(plan, as_torch(tensors), x)
        hc = HipCircuit(plan, tensors, device=hip_device)
        assert hc._pad_info is not None
        got = hc(x.to(hip_device)).cpu()
        assert torch.isfinite(got).all(), kw
        assert float((got - want).abs().max()) <= REL * float(want.abs().max()), kw


@pytest.mark.parametrize("name", ["cfg2_qt784", "quadgraph_6x6_k4", "cfg4_pd784"])
def test_graph_replay_equals_eager_replay(hip_device, name):
    """The recorded launch list replayed as a hipGraph (`graph_min_launches=0`) and eagerly by the native executor
    (the default for lists this short) give bit-identical outputs, forward and log-likelihood sum."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case(name)
    x = _x_of(plan, g).to(hip_device)
    eager = HipCircuit(plan, tensors, device=hip_device)
    graph = HipCircuit(plan, tensors, device=hip_device, graph_min_launches=0)
    assert not eager.replays_as_graph(x.shape[0]) and graph.replays_as_graph(x.shape[0])
    for _ in range(2):  # second call: the instantiated graph is replayed
        assert torch.equal(eager(x), graph(x))
    assert torch.equal(eager.log_likelihood_sum(x), graph.log_likelihood_sum(x))


@pytest.mark.parametrize("B", [1, 15, 16, 17, 31, 32, 33, 100, 255, 257, 511, 2047, 4097])
def test_default_path_over_batch_sizes(hip_device, B):
    """The default evaluation of the north-star circuit (leaf launch on the raw batch, tail + next parameters in one launch)
    at ragged batch sizes: bit-identical to the three-launch form with a staged batch, within fp32 rounding of the
    layer-by-layer evaluation, LL sum = sum of the outputs; a second batch right behind the first (the parameters it uses
    were evaluated by the first one's last launch)."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    a = HipCircuit(plan, tensors, device=hip_device)
    b = HipCircuit(plan, tensors, device=hip_device, params_at_end=False, direct_input=False)
    c = HipCircuit(plan, tensors, device=hip_device, fuse=False, use_graph=False)
    gen = torch.Generator().manual_seed(1000 + B)
    for rep in range(2):
        x = torch.randint(0, 256, (B, 784), generator=gen)
        if rep == 1 and B > 2:
            x[B // 2, ::3] = -1  # (marginalised variables)
        x = x.to(hip_device)
        ya, yb, yc = a(x).clone(), b(x).clone(), c(x).clone()
        assert torch.equal(ya, yb), (B, rep, float((ya - yb).abs().max()))
        assert float(((ya - yc).abs() / yc.abs()).max()) <= 2e-6, (B, rep)
        ll = a.log_likelihood_sum(x).cpu()
        assert ll[1].item() == B and abs(ll[0].item() - float(ya.double().sum())) <= 1e-6 * abs(ll[0].item())


def test_log_likelihood_sum_into_a_row_of_the_callers_buffer(hip_device):
    """`log_likelihood_sum(x, out=row)`: the pair of each step lands in the caller's (steps, 2) buffer -- written by the
    forward's last launch on the default path (ck_tail_params_fwd's ll_cell), by a 16-byte copy on the others -- and equals
    what the call without `out` returns; the binding's own pair is written again as soon as `out` is left out."""
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, g = load_case("cfg2_qt784")
    gen = torch.Generator().manual_seed(77)
    xs = [torch.randint(0, 256, (64, 784), generator=gen).to(hip_device) for _ in range(5)]
    for kw in ({}, {"params_at_end": False}, {"fuse": False, "use_graph": False}):
        hc = HipCircuit(plan, tensors, device=hip_device, **kw)
        want = torch.stack([hc.log_likelihood_sum(x).clone() for x in xs])
        assert (hc._bind(64).ll_cell == 1) == (kw == {})
        buf = torch.full((5, 2), float("nan"), dtype=torch.float64, device=hip_device)
        own = hc._bind(64).ll
        own.fill_(-1.0)
        for k, x in enumerate(xs):
            r = hc.log_likelihood_sum(x, out=buf[k])
            assert r.data_ptr() == buf[k].data_ptr()
        assert torch.equal(buf, want), kw
        if kw == {}:
            assert torch.equal(own.cpu(), torch.tensor([-1.0, -1.0], dtype=torch.float64))  # (nothing was written there)
        assert torch.equal(hc.log_likelihood_sum(xs[2]), want[2]) and hc.log_likelihood_sum(xs[2]).data_ptr() == own.data_ptr()
        with pytest.raises(ValueError, match="float64"):
            hc.log_likelihood_sum(xs[0], out=torch.zeros(2, device=hip_device))
        with pytest.raises(ValueError, match="float64"):
            hc.log_likelihood_sum(xs[0], out=torch.zeros((2, 2), dtype=torch.float64, device=hip_device)[:, 0])
