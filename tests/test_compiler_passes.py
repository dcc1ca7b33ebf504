"""`fuse_plan` / `fold_plan` (cirkit_amd/compiler.py) against the reference's own post-processing:
the reference's known-answer circuits were compiled by the REAL reference under all four
fold/optimize flag combinations (tests/golden/kat_*_f?o?); applying our passes to the
unfolded / unoptimised plan must reproduce the reference's optimised / folded plans exactly --
layers, fold order, index arrays AND the stacked literal weights."""
import numpy as np
import pytest

from conftest import load_case
from cirkit_amd.compiler import compile_plan, fold_plan, fuse_plan


def _same(plan_a, vals_a, plan_b, vals_b):
    da, aa = plan_a.to_json()
    db, ab = plan_b.to_json()
    da["name"] = db["name"] = ""
    for i, (x, y) in enumerate(zip(da["layers"], db["layers"])):
        assert x == y, f"layer {i}: {x} != {y}"
    assert da == db
    assert set(aa) == set(ab)
    for k in aa:
        assert np.array_equal(aa[k], ab[k]), k
    assert set(vals_a) == set(vals_b)
    for k in vals_a:
        assert np.array_equal(np.asarray(vals_a[k]), np.asarray(vals_b[k])), k


@pytest.mark.parametrize("fam", ["bernoulli", "gaussian"])
def test_fuse_matches_reference_optimize(fam):
    p00, t00, _ = load_case(f"kat_{fam}_f0o0")
    p01, t01, _ = load_case(f"kat_{fam}_f0o1")
    _same(*fuse_plan(p00, t00), p01, t01)


@pytest.mark.parametrize("fam", ["bernoulli", "gaussian"])
@pytest.mark.parametrize("opt", [0, 1])
def test_fold_matches_reference_fold(fam, opt):
    pu, tu, _ = load_case(f"kat_{fam}_f0o{opt}")
    pf, tf, _ = load_case(f"kat_{fam}_f1o{opt}")
    _same(*fold_plan(pu, tu), pf, tf)


@pytest.mark.parametrize("fam", ["bernoulli", "gaussian"])
def test_full_pipeline_and_function_preservation(fam):
    from oracle.torch_oracle import as_torch, evaluate_plan
    import torch

    p00, t00, g = load_case(f"kat_{fam}_f0o0")
    p11, t11, _ = load_case(f"kat_{fam}_f1o1")
    plan, vals = compile_plan(p00, t00)
    _same(plan, vals, p11, t11)
    x = torch.from_numpy(g["x"])
    y = evaluate_plan(plan, as_torch(vals), x)
    assert np.array_equal(y.numpy(), evaluate_plan(p11, as_torch(t11), x).numpy())
    assert float((y.double() - torch.from_numpy(g["y_f32"]).double()).abs().max()) < 1e-5


def test_fold_rejects_folded_input():
    p11, t11, _ = load_case("kat_bernoulli_f1o1")
    with pytest.raises(ValueError):
        fold_plan(p11, t11)
