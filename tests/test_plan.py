import json

import numpy as np
import pytest
import torch

from conftest import load_case
from cirkit_amd.initializers import init_plan_tensors, pseudo_normal, stream_of
from cirkit_amd.plan import IDX_ARRAY, IDX_NONE, IDX_UNSQ0, IDX_UNSQ1, FoldIndex, Plan, resolve_fold_index


@pytest.mark.parametrize("name", ["cfg1_rbt8", "cfg2_qt784", "cfg4_pd784", "cfg5_sos_z_k32"])
def test_json_round_trip(name):
    plan, _, _ = load_case(name)
    doc, arrays = plan.to_json()
    again = Plan.from_json(json.loads(json.dumps(doc)), arrays)
    doc2, arrays2 = again.to_json()
    assert doc == doc2
    assert set(arrays) == set(arrays2)
    for k in arrays:
        assert np.array_equal(arrays[k], arrays2[k])


def test_plan_version_is_checked():
    plan, _, _ = load_case("cfg1_rbt8")
    doc, arrays = plan.to_json()
    doc["version"] = 999
    with pytest.raises(ValueError):
        Plan.from_json(doc, arrays)


def test_config2_shape_matches_the_survey():
    """12 folded layers, fold counts, 8 026 144 parameters."""
    plan, _, _ = load_case("cfg2_qt784")
    assert [l.num_folds for l in plan.layers] == [784, 784, 392, 196, 98, 49, 24, 11, 6, 4, 2, 1]
    assert [l.type for l in plan.layers] == ["categorical", "sum"] + ["cpt"] * 10
    assert plan.num_params == 8026144
    assert plan.layers[7].inputs.ids == [6, 5]
    assert plan.layers[1].inputs.kind == IDX_UNSQ1 and plan.layers[11].inputs.kind == IDX_UNSQ0
    assert np.array_equal(plan.layers[0].scope_idx[:, 0], np.arange(784))


def test_algorithmic_bytes_of_config2():
    """2522.0 MB per 4096-batch = 615.7 KB per evaluation."""
    plan, _, _ = load_case("cfg2_qt784")
    a = plan.algorithmic_bytes(4096)
    assert a["total"] == 2521964672.0
    assert round(a["read"] / 1e6, 1) == 1257.8 and round(a["write"] / 1e6, 1) == 1232.1
    assert round(a["total"] / 4096 / 1e3, 1) == 615.7


@pytest.mark.parametrize("kind", [IDX_ARRAY, IDX_UNSQ0, IDX_UNSQ1, IDX_NONE])
def test_resolve_fold_index_matches_cat_then_index(kind):
    """Same semantics as `torch.cat(outs)[idx]` (circuits.py:42-47 of the reference)."""
    rng = np.random.default_rng(0)
    folds = {0: 3, 1: 5, 2: 2}
    outs = {i: torch.arange(n, dtype=torch.float32) + 100 * i for i, n in folds.items()}
    ids = [2, 0, 1]
    cat = torch.cat([outs[i] for i in ids])
    if kind == IDX_ARRAY:
        arr = rng.integers(0, len(cat), size=(4, 3))
        fi = FoldIndex(ids, kind, arr)
        want = cat[torch.from_numpy(arr)]
    elif kind == IDX_UNSQ0:
        fi, want = FoldIndex(ids, kind), cat[None]
    elif kind == IDX_UNSQ1:
        fi, want = FoldIndex(ids, kind), cat[:, None]
    else:
        fi, want = FoldIndex(ids, kind), cat
    pairs = resolve_fold_index(fi, [folds[i] for i in range(3)])
    got = torch.tensor([[float(outs[int(p)][int(f)]) for p, f in row] for row in pairs.reshape(-1, 1, 2)]).reshape(want.shape)
    assert torch.equal(got, want)


def test_resolve_fold_index_rejects_out_of_range():
    with pytest.raises(ValueError):
        resolve_fold_index(FoldIndex([0], IDX_ARRAY, np.array([[0, 7]])), [3])


def test_closed_form_parameters_are_reproducible():
    a = pseudo_normal((4, 5), stream=7, seed=0)
    b = pseudo_normal((4, 5), stream=7, seed=0)
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert not np.array_equal(a, pseudo_normal((4, 5), stream=8, seed=0))
    big = pseudo_normal((200000,), stream=stream_of("t0"))
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01
    assert float(np.abs(big).max()) <= 2 * 1.7320508075688772 + 1e-6
    # pinned values: any change to the generator invalidates every golden fixture
    assert np.allclose(pseudo_normal((3,), stream=1, seed=0), [-0.22580937, 0.6800974, 0.36141655], atol=1e-7)
    plan, _, _ = load_case("cfg1_rbt8")
    t = init_plan_tensors(plan)
    assert set(t) == set(plan.tensors) and all(v.shape == tuple(plan.tensors[k][0]) for k, v in t.items())
