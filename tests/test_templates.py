"""The native plan builders (cirkit_amd/templates.py) must emit THE SAME folded plan as the
reference's ``image_data/tabular_data -> compile(fold=True, optimize=True)`` (SURVEY.md section 8 c:
"pinned by the plan fixtures"): same layer list, fold order, index arrays, scope order and
parameter-tensor order -- compared field by field against plans extracted from the real reference
(tests/golden/make_fixtures.py)."""
import glob
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN
from cirkit_amd.plan import Plan
from cirkit_amd.templates import (
    InputSpec,
    _kahn,
    _kahn_frontiers,
    quad_tree,
    quad_tree_plan,
    random_binary_tree,
    random_binary_tree_plan,
)


def _assert_same_plan(built: Plan, ref: Plan):
    da, aa = built.to_json()
    db, ab = ref.to_json()
    da["name"] = db["name"] = ""
    assert len(da["layers"]) == len(db["layers"])
    for i, (x, y) in enumerate(zip(da["layers"], db["layers"])):
        assert x == y, f"layer {i} differs"
    assert da == db
    assert set(aa) == set(ab)
    for k in aa:
        assert np.array_equal(aa[k], ab[k]), k


def test_quadtree_784_cp_matches_reference_plan():
    built = quad_tree_plan((1, 28, 28), input_layer=InputSpec("categorical", 256), sum_product="cp",
                           num_input_units=32, num_sum_units=32)
    _assert_same_plan(built, Plan.load(os.path.join(GOLDEN, "cfg2_qt784")))
    assert built.algorithmic_bytes(4096)["total"] == 2521964672.0


def test_quadtree_784_cpt_and_embedding_variants_match():
    built = quad_tree_plan((1, 28, 28), input_layer=InputSpec("categorical", 256), sum_product="cp-t",
                           num_input_units=16, num_sum_units=16)
    _assert_same_plan(built, Plan.load(os.path.join(GOLDEN, "cfg2t_qt784_cpt16")))
    built = quad_tree_plan((1, 28, 28), input_layer=InputSpec("embedding", 256), sum_product="cp-t",
                           num_input_units=32, num_sum_units=32, sum_activation="none", semiring="complex-lse-sum")
    _assert_same_plan(built, Plan.load(os.path.join(GOLDEN, "cfg5_sos_c_k32")))


def test_random_binary_tree_8_matches_reference_plan():
    built = random_binary_tree_plan(8, input_layer=InputSpec("categorical", 4), sum_product="cp",
                                    num_input_units=4, num_sum_units=4)
    _assert_same_plan(built, Plan.load(os.path.join(GOLDEN, "cfg1_rbt8")))


PLAN_ONLY = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLDEN, "plan_*.json"))
                   if not os.path.basename(p).startswith("plan_clt_"))  # (structures the reference learned from data: parity fixtures only)


def _rebuild(name: str) -> Plan:
    """The native call that corresponds to the reference call which produced fixture `name`
    (tests/golden/make_fixtures.py:plans_only)."""
    from cirkit_amd.templates import build_plan, fully_factorized, image_data, linear_tree, poon_domingos, tabular_data

    sps = {"cp": "cp", "cpt": "cp-t", "tucker": "tucker"}
    m = re.fullmatch(r"plan_quadtree(\d)_(\d+)x(\d+)_(cp|cpt)", name)
    if m:
        return quad_tree_plan((1, int(m[2]), int(m[3])), num_patch_splits=int(m[1]), input_layer=InputSpec("categorical", 256),
                              sum_product=sps[m[4]], num_input_units=3, num_sum_units=3)
    m = re.fullmatch(r"plan_rbt(\d+)_dNone_(cp|cpt)", name)
    if m:
        return random_binary_tree_plan(int(m[1]), input_layer=InputSpec("categorical", 3), sum_product=sps[m[2]],
                                       num_input_units=2, num_sum_units=2)
    m = re.fullmatch(r"plan_(quadgraph|quadtree2|quadtree4|poondomingos|randombinarytree)_(\d+)x(\d+)x(\d+)_(cp|cpt|tucker)(_mixFalse)?(_nc3)?", name)
    if m:
        rg = {"quadgraph": "quad-graph", "quadtree2": "quad-tree-2", "quadtree4": "quad-tree-4", "poondomingos": "poon-domingos",
              "randombinarytree": "random-binary-tree"}[m[1]]
        return image_data((int(m[2]), int(m[3]), int(m[4])), rg, input_layer="categorical", num_input_units=3,
                          sum_product_layer=sps[m[5]], num_sum_units=3, use_mixing_weights=not m[6], num_classes=3 if m[7] else 1)
    small = dict(num_input_units=2, num_sum_units=2, sum_product="cp")
    if name == "plan_pd_1x12x12_delta6-3_cp":
        return build_plan(poon_domingos((1, 12, 12), delta=[6, 3]), input_layer=InputSpec("categorical", 5), **small)
    if name == "plan_pd_1x9x6_delta2_depth2_cp":
        return build_plan(poon_domingos((1, 9, 6), delta=2, max_depth=2), input_layer=InputSpec("categorical", 5), **small)
    if name == "plan_rbt11_d2_cp":
        return random_binary_tree_plan(11, depth=2, input_layer=InputSpec("categorical", 3), **small)
    if name == "plan_rbt19_d3_cpt":
        return random_binary_tree_plan(19, depth=3, seed=7, input_layer=InputSpec("categorical", 3), **{**small, "sum_product": "cp-t"})
    cat3 = dict(input_layer=InputSpec("categorical", 3), **small)
    if name == "plan_ff5_r1_cp":
        return build_plan(fully_factorized(5), **cat3)
    if name == "plan_ff4_r3_cp":
        return build_plan(fully_factorized(4, num_repetitions=3), **cat3)
    if name == "plan_lt6_r1_cp":
        return build_plan(linear_tree(6), **cat3)
    if name == "plan_lt5_r2_rand3_cpt":
        return build_plan(linear_tree(5, num_repetitions=2, randomize=True, seed=3), **{**cat3, "sum_product": "cp-t"})
    if name == "plan_lt4_order2031_cp":
        return build_plan(linear_tree(4, ordering=[2, 0, 3, 1]), **cat3)
    if name == "plan_rbt6_perfeature_cp":
        return tabular_data("random-binary-tree", num_features=6,
                            input_layers=[{"name": "categorical", "args": {"num_categories": 3}}, {"name": "gaussian", "args": {}}] * 3,
                            num_input_units=2, sum_product_layer="cp", num_sum_units=2)
    raise AssertionError(f"no native recipe for fixture {name}")


@pytest.mark.parametrize("name", PLAN_ONLY)
def test_awkward_shapes_match_reference_plans(name):
    _assert_same_plan(_rebuild(name), Plan.load(os.path.join(GOLDEN, name)))


def test_poon_domingos_784_gaussian_matches_reference_plan():
    """BASELINE config 4: 10 904 symbolic layers -> 38 folded, mixing layers of arity 2..12, one
    collapsed Sum -> Sum pair whose weight is a MatMul parameter node."""
    from cirkit_amd.templates import image_data

    built = image_data((1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64,
                       sum_product_layer="cp", num_sum_units=64)
    _assert_same_plan(built, Plan.load(os.path.join(GOLDEN, "cfg4_pd784")))
    assert [n.op for n in built.layers[35].params["weight"].nodes] == ["tensor", "tensor", "softmax", "softmax", "mixing_weight", "matmul"]


def test_template_entry_points_mirror_the_reference_signatures():
    from cirkit_amd.templates import image_data, tabular_data

    a = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32,
                   sum_product_layer="cp", num_sum_units=32)
    _assert_same_plan(a, Plan.load(os.path.join(GOLDEN, "cfg2_qt784")))
    b = tabular_data("random-binary-tree", num_features=8,
                     input_layers={"name": "categorical", "args": {"num_categories": 4}},
                     num_input_units=4, sum_product_layer="cp", num_sum_units=4)
    _assert_same_plan(b, Plan.load(os.path.join(GOLDEN, "cfg1_rbt8")))
    with pytest.raises(ValueError):
        image_data((1, 28, 28), region_graph="hexagons", num_input_units=4, num_sum_units=4)
    assert image_data((1, 4, 4), input_layer="binomial", num_input_units=4, num_sum_units=4).layers[0].config["total_count"] == 255
    with pytest.raises(NotImplementedError):  # structure learning is out of scope: compile with cirkit instead
        tabular_data("chow-liu-tree", num_features=4, input_layers={"name": "gaussian", "args": {}},
                     num_input_units=2, num_sum_units=2)


def test_region_graph_structure():
    rg = quad_tree(28, 28)
    regions = [n for n in rg.nodes if rg.is_region[n]]
    parts = [n for n in rg.nodes if not rg.is_region[n]]
    assert len([r for r in regions if not rg.ins.get(r)]) == 784
    assert rg.scope[rg.root] == tuple(range(784))
    for p in parts:  # every partition splits its scope exactly
        kids = rg.ins[p]
        assert sorted(v for k in kids for v in rg.scope[k]) == list(rg.scope[p])
    rb = random_binary_tree(8)
    leaves = [n for n in rb.nodes if rb.is_region[n] and not rb.ins.get(n)]
    assert sorted(v for l in leaves for v in rb.scope[l]) == list(range(8))
    with pytest.raises(ValueError):
        quad_tree(0, 3)
    with pytest.raises(ValueError):
        random_binary_tree(8, depth=9)


def test_orderings():
    ins = {"c": ["a", "b"], "d": ["c", "a"], "e": ["b"]}
    nodes = ["a", "b", "c", "d", "e"]
    assert _kahn(nodes, ins) == ["a", "b", "c", "e", "d"]
    assert _kahn_frontiers(nodes, ins) == [["a", "b"], ["c", "e"], ["d"]]
    with pytest.raises(ValueError):
        _kahn(["x", "y"], {"x": ["y"], "y": ["x"]})


@pytest.mark.gpu
def test_native_plan_evaluates_like_the_reference(hip_device):
    """End to end without any fixture plan: build natively, evaluate on the GPU, compare with the
    reference's outputs for the same closed-form parameters."""
    import torch

    from conftest import load_case
    from cirkit_amd.circuit import HipCircuit
    from cirkit_amd.initializers import init_plan_tensors

    _, _, g = load_case("cfg2_qt784")
    plan = quad_tree_plan((1, 28, 28), input_layer=InputSpec("categorical", 256), sum_product="cp",
                          num_input_units=32, num_sum_units=32)
    hc = HipCircuit(plan, init_plan_tensors(plan), device=hip_device)
    y = hc(torch.from_numpy(g["x"].astype(np.int64)).to(hip_device)).cpu()
    ref = torch.from_numpy(g["y_f32"])
    assert float(((y - ref).abs() / ref.abs()).max()) <= 1e-4


