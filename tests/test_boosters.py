"""XGBoost base learner of the stack (f4): UBJSON model reader, tree flattening, and the GPU predict kernel against the numpy
restatement of XGBoost's predict rule (oracle/reference_cpu.py: xgb_predict).  PARITY UNPINNED: the xgboost package is not in the
build image, so nothing here is compared with the library's own ``predict``; the fixture tests/golden/xgb_maccs_head.npz holds
the first trees of the model the reference ships (Models/xgb_model_maccs.pkl), lifted by tools/make_golden.py."""
import os
import struct

import numpy as np
import pytest
import torch

from bbbp_amd import boosters
from oracle import reference_cpu as oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xgb_maccs_head.npz")


# ---- a minimal UBJSON writer laid out like XGBoost's (typed arrays "[$d#L<n>", int64 lengths, string-valued parameters) ----
def _i64(n):
    return b"L" + struct.pack(">q", n)


def _key(k):
    return _i64(len(k)) + k.encode()


def ubj(v):
    if isinstance(v, dict):
        return b"{" + b"".join(_key(k) + ubj(x) for k, x in v.items()) + b"}"
    if isinstance(v, np.ndarray):
        tag = {"float32": "d", "int32": "l", "uint8": "U", "int64": "L"}[str(v.dtype)]
        return b"[$" + tag.encode() + b"#" + _i64(len(v)) + v.astype(">" + v.dtype.str[1:]).tobytes()
    if isinstance(v, list):
        return b"[#" + _i64(len(v)) + b"".join(ubj(x) for x in v)
    if isinstance(v, str):
        return b"S" + _i64(len(v)) + v.encode()
    if isinstance(v, bool):
        return b"T" if v else b"F"
    if isinstance(v, int):
        return b"l" + struct.pack(">i", v)
    if isinstance(v, float):
        return b"D" + struct.pack(">d", v)
    if v is None:
        return b"Z"
    raise TypeError(type(v))


def random_tree(rng, n_features, depth, p_leaf=0.25):
    """One regression tree in XGBoost's array form (breadth-first node ids, leaves carry their weight in split_conditions)."""
    left, right, feat, cond, dl = [], [], [], [], []
    frontier = [(0, 0)]
    left.append(-1); right.append(-1); feat.append(0); cond.append(0.0); dl.append(0)
    while frontier:
        node, d = frontier.pop(0)
        if d < depth and (d < 2 or rng.random() > p_leaf):
            l, r = len(left), len(left) + 1
            for _ in range(2):
                left.append(-1); right.append(-1); feat.append(0); cond.append(0.0); dl.append(0)
            left[node], right[node] = l, r
            feat[node] = int(rng.integers(0, n_features))
            cond[node] = float(np.float32(rng.normal()))
            dl[node] = int(rng.integers(0, 2))
            frontier += [(l, d + 1), (r, d + 1)]
        else:
            cond[node] = float(np.float32(0.1 * rng.normal()))
    return dict(left_children=np.array(left, np.int32), right_children=np.array(right, np.int32), split_indices=np.array(feat, np.int32),
                split_conditions=np.array(cond, np.float32), default_left=np.array(dl, np.uint8), split_type=np.zeros(len(left), np.uint8),
                tree_param=dict(num_nodes=str(len(left)), num_feature=str(n_features)))


def random_model(seed, n_trees, n_features, depth, base="-5.536508E-2"):
    rng = np.random.default_rng(seed)
    trees = [random_tree(rng, n_features, depth) for _ in range(n_trees)]
    return {"Config": {"learner": {"generic_param": {"device": "cpu"}}},
            "Model": {"learner": {"attributes": {}, "feature_names": [], "gradient_booster": {"model": {
                "gbtree_model_param": {"num_parallel_tree": "1", "num_trees": str(n_trees)}, "trees": trees,
                "tree_info": np.zeros(n_trees, np.int32)}, "name": "gbtree"},
                "learner_model_param": {"base_score": base, "boost_from_average": "1", "num_class": "0", "num_feature": str(n_features), "num_target": "1"},
                "objective": {"name": "reg:squarederror", "reg_loss_param": {"scale_pos_weight": "1"}}}, "version": [2, 0, 3]}}


def test_ubjson_reader_on_the_library_writers_own_bytes():
    """Tree 0 of the shipped model, as the library wrote it: typed big-endian arrays, int64 lengths, string parameters."""
    g = np.load(GOLD)
    tree = boosters.parse_ubjson(g["tree0_ubjson"].tobytes())
    n0 = int(g["root"][1])
    assert tree["tree_param"]["num_nodes"] == str(n0) and tree["tree_param"]["num_feature"] == "49319" and tree["id"] == 0
    assert tree["left_children"].dtype == np.int32 and tree["split_conditions"].dtype == np.float32 and tree["default_left"].dtype == np.uint8
    want_left = np.where(g["left"][:n0] >= 0, g["left"][:n0], -1)
    assert np.array_equal(tree["left_children"], want_left) and np.array_equal(tree["split_indices"], g["feature"][:n0])
    assert np.array_equal(tree["split_conditions"], g["cond"][:n0]) and tree["parents"][0] == 2147483647
    assert int(g["n_trees_total"]) == 300 and int(g["n_features"]) == 49319 and abs(float(g["base_score"]) + 0.05536508) < 1e-7


def test_ubjson_round_trip_and_flatten():
    doc = random_model(3, n_trees=5, n_features=40, depth=6)
    back = boosters.parse_ubjson(ubj(doc))
    t0, b0 = doc["Model"]["learner"]["gradient_booster"]["model"]["trees"][2], back["Model"]["learner"]["gradient_booster"]["model"]["trees"][2]
    for k in ("left_children", "right_children", "split_indices", "split_conditions", "default_left"):
        assert np.array_equal(t0[k], b0[k]) and t0[k].dtype == b0[k].dtype
    assert back["Model"]["version"] == [2, 0, 3] and back["Config"]["learner"]["generic_param"]["device"] == "cpu"
    left, right, feature, cond, dleft, root, nf, base = boosters.XGBTrees.flatten(back)
    assert nf == 40 and abs(base + 0.05536508) < 1e-12 and len(root) == 6 and root[-1] == len(left)
    # rebased children stay inside their own tree
    for t in range(5):
        sl = slice(root[t], root[t + 1])
        inner = left[sl] >= 0
        assert (left[sl][inner] > root[t]).all() and (right[sl][inner] < root[t + 1]).all()
    # other scalar types and unsized containers of the specification
    blob = b"{" + b"i\x01a" + b"[" + b"i\x05" + b"U\xff" + b"I\x01\x00" + b"d" + struct.pack(">f", 1.5) + b"T" + b"F" + b"Z" + b"Ci" + b"]" + b"}"
    assert boosters.parse_ubjson(blob) == {"a": [5, 255, 256, 1.5, True, False, None, "i"]}
    with pytest.raises(ValueError):
        boosters.parse_ubjson(b"{i\x01a?}")


def test_unsupported_models_are_refused():
    doc = random_model(4, 2, 10, 3)
    doc["Model"]["learner"]["objective"]["name"] = "binary:logistic"
    with pytest.raises(ValueError):
        boosters.XGBTrees.flatten(doc)
    doc = random_model(4, 2, 10, 3)
    doc["Model"]["learner"]["gradient_booster"]["model"]["trees"][0]["split_type"][0] = 1
    with pytest.raises(ValueError):
        boosters.XGBTrees.flatten(doc)
    m = boosters.XGBTrees(*boosters.XGBTrees.flatten(random_model(4, 2, 10, 3)), device="cpu")
    with pytest.raises(RuntimeError):
        m.predict(np.zeros((3, 10), np.float32))          # no CPU fallback


def test_oracle_rule_on_a_hand_case():
    """x < condition goes left (equality goes right), NaN takes the default child, float32 sum on top of base_score."""
    left = np.array([1, -1, -1, 4, -1, -1]); right = np.array([2, -1, -1, 5, -1, -1])
    feature = np.array([0, 0, 0, 1, 0, 0]); cond = np.array([0.5, -1.0, 2.0, 0.0, 10.0, 20.0], np.float32)
    dleft = np.array([1, 0, 0, 0, 0, 0], np.uint8); root = np.array([0, 3, 6])
    X = np.array([[0.4, -1.0], [0.5, 0.0], [np.nan, np.nan], [0.6, 1.0]], np.float32)
    got = oracle.xgb_predict(left, right, feature, cond, dleft, root, 0.25, X)
    assert np.array_equal(got, np.float32(0.25) + np.array([-1.0 + 10.0, 2.0 + 20.0, -1.0 + 20.0, 2.0 + 20.0], np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_trees,depth,nan_frac", [(1, 3, 4, 0.0), (700, 40, 12, 0.1), (5000, 300, 16, 0.02)])
def test_gpu_predict_is_bit_exact_against_the_oracle(n, n_trees, depth, nan_frac):
    doc = random_model(100 + n, n_trees, 64, depth)
    m = boosters.XGBTrees.from_raw(ubj(doc))
    rng = np.random.default_rng(n)
    X = rng.normal(size=(n, 64)).astype(np.float32)
    X[rng.random(X.shape) < nan_frac] = np.nan
    # some rows sit exactly on thresholds: equality must go right
    a = m.arrays
    inner = np.flatnonzero(a["left"] >= 0)[: min(n, 50)]
    for i, node in enumerate(inner):
        X[i % n, a["feature"][node]] = a["cond"][node]
    got = m.predict(X)
    want = oracle.xgb_predict(a["left"], a["right"], a["feature"], a["cond"], a["default_left"], a["root"], m.base_score, X)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    # chunked calls give the same bits
    assert np.array_equal(m.predict_device(torch.from_numpy(X), rows_per_call=257).cpu().numpy(), want)


@pytest.mark.gpu
def test_gpu_predict_on_the_shipped_models_first_trees():
    g = np.load(GOLD)
    m = boosters.XGBTrees(g["left"], g["right"], g["feature"], g["cond"], g["default_left"], g["root"], int(g["n_features"]), float(g["base_score"]))
    rng = np.random.default_rng(9)
    X = rng.random((333, m.n_features), dtype=np.float32)          # standardised MACCS bits + pixels live around [0, 1]
    want = oracle.xgb_predict(g["left"], g["right"], g["feature"], g["cond"], g["default_left"], g["root"], float(g["base_score"]), X)
    assert np.array_equal(m.predict(X), want) and len(np.unique(want)) > 50


# ---- CatBoost (oblivious trees, JSON export) -------------------------------------------------------------------------------------
def random_catboost_doc(seed, n_trees, n_features, depth, nan_true_every=7):
    rng = np.random.default_rng(seed)
    feats = [dict(feature_index=i, flat_feature_index=i, borders=[], has_nans=bool(i % 3 == 0),
                  nan_value_treatment="AsTrue" if i % nan_true_every == 0 else ("AsFalse" if i % 2 else "AsIs")) for i in range(n_features)]
    trees = []
    for t in range(n_trees):
        d = int(rng.integers(1, depth + 1)) if t else depth
        splits = [dict(float_feature_index=int(rng.integers(0, n_features)), split_index=int(i), split_type="FloatFeature",
                       border=float(np.float32(rng.normal()))) for i in range(d)]
        trees.append(dict(splits=splits, leaf_values=[float(v) for v in 0.1 * rng.normal(size=1 << d)], leaf_weights=[1.0] * (1 << d)))
    return dict(model_info={}, features_info=dict(float_features=feats), oblivious_trees=trees, scale_and_bias=[0.75, [-0.3]])


def test_catboost_flatten_and_oracle_rule():
    import json
    doc = random_catboost_doc(1, n_trees=4, n_features=12, depth=5)
    sf, sb, nt, fs, fl, lv, nf, scale, bias = boosters.CatBoostTrees.flatten(json.loads(json.dumps(doc)))
    assert nf == 12 and scale == 0.75 and bias == -0.3 and len(fs) == 5 and fs[1] == 5 and len(fl) == 4 and lv.size == sum(1 << (fs[i + 1] - fs[i]) for i in range(4))
    assert nt[0] == 1 and nt[7] == 1 and nt[1] == 0
    # hand case: one tree of depth 2 over features 0 and 1; bit 0 = x0 > 0.5, bit 1 = x1 > -1
    hand = dict(features_info=dict(float_features=[dict(feature_index=0, flat_feature_index=0), dict(feature_index=1, flat_feature_index=1, nan_value_treatment="AsTrue")]),
                oblivious_trees=[dict(splits=[dict(float_feature_index=0, border=0.5), dict(float_feature_index=1, border=-1.0)], leaf_values=[1.0, 2.0, 4.0, 8.0])],
                scale_and_bias=[2.0, [0.5]])
    f = boosters.CatBoostTrees.flatten(hand)
    X = np.array([[0.4, -2.0], [0.6, -2.0], [0.5, 0.0], [np.nan, np.nan]], np.float32)
    got = oracle.catboost_predict(*f[:6], f[7], f[8], X)
    assert np.array_equal(got, 2.0 * np.array([1.0, 2.0, 4.0, 4.0]) + 0.5)       # equality is not '>', NaN: false on x0, true on x1
    bad = json.loads(json.dumps(doc)); bad["oblivious_trees"][0]["splits"][0]["split_type"] = "OneHotFeature"
    with pytest.raises(ValueError):
        boosters.CatBoostTrees.flatten(bad)
    with pytest.raises(RuntimeError):
        boosters.CatBoostTrees(*f, device="cpu").predict(X)


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_trees,depth", [(1, 2, 3), (777, 50, 6), (4000, 300, 10)])
def test_gpu_catboost_predict_is_bit_exact_against_the_oracle(n, n_trees, depth):
    doc = random_catboost_doc(n, n_trees, 96, depth)
    m = boosters.CatBoostTrees.from_json(doc)
    rng = np.random.default_rng(n + 1)
    X = rng.normal(size=(n, 96)).astype(np.float32)
    X[rng.random(X.shape) < 0.05] = np.nan
    a = m.arrays
    for i in range(min(n, 40)):                      # rows sitting exactly on a border: '>' is false
        s = i % len(a["split_feature"])
        X[i, a["split_feature"][s]] = a["split_border"][s]
    want = oracle.catboost_predict(a["split_feature"], a["split_border"], a["nan_true"], a["tree_first_split"], a["tree_first_leaf"], a["leaf_values"],
                                   m.scale, m.bias, X)
    got = m.predict(X)
    assert got.dtype == np.float64 and np.array_equal(got, want)
    assert np.array_equal(m.predict_device(torch.from_numpy(X), rows_per_call=301).cpu().numpy(), want)


# ---- untrusted model files: corrupted arrays must be refused on the host, before any array reaches the GPU --------
def _flat(seed=11):
    return list(boosters.XGBTrees.flatten(random_model(seed, n_trees=3, n_features=20, depth=5)))


@pytest.mark.parametrize("corrupt", ["feature_too_large", "feature_negative", "child_outside_tree", "cycle", "two_parents", "half_leaf",
                                     "length_mismatch", "root_is_child", "bad_offsets"])
def test_corrupted_gbtree_arrays_are_refused(corrupt):
    left, right, feature, cond, dleft, root, nf, base = _flat()
    left, right, feature = left.copy(), right.copy(), feature.copy()
    inner = np.flatnonzero(left >= 0)
    node = int(inner[1])
    if corrupt == "feature_too_large":
        feature[node] = nf
    elif corrupt == "feature_negative":
        feature[node] = -3
    elif corrupt == "child_outside_tree":
        left[node] = int(root[1]) + 1 if node < root[1] else 0          # a node of another tree
    elif corrupt == "cycle":
        left[int(left[node])] = node; right[int(left[node])] = node      # the child points back at its parent
    elif corrupt == "two_parents":
        right[node] = left[node]
    elif corrupt == "half_leaf":
        right[node] = -1
    elif corrupt == "length_mismatch":
        cond = cond[:-1]
    elif corrupt == "root_is_child":
        left[node] = int(root[np.searchsorted(root, node, side="right") - 1])
    elif corrupt == "bad_offsets":
        root = root.copy(); root[1] = root[2]
    with pytest.raises(ValueError):
        boosters.validate_gbt(left, right, feature, cond, dleft, root, nf)
    with pytest.raises(ValueError):
        boosters.XGBTrees(left, right, feature, cond, dleft, root, nf, base, device="cpu")


def test_valid_gbtree_arrays_pass_and_report_their_depth():
    left, right, feature, cond, dleft, root, nf, base = _flat()
    assert 2 <= boosters.validate_gbt(left, right, feature, cond, dleft, root, nf) <= 5
    g = np.load(GOLD)                                       # the shipped model's first trees
    assert 1 <= boosters.validate_gbt(g["left"], g["right"], g["feature"], g["cond"], g["default_left"], g["root"], int(g["n_features"])) <= 30
    # a chain deeper than the device walk's bound is refused
    n = boosters.GBT_MAX_DEPTH + 2
    left = np.arange(1, 2 * n + 1, 2); right = left + 1                  # node i -> children 2i+1 (inner), 2i+2 (leaf) laid out pairwise
    L = np.full(2 * n + 1, -1); R = np.full(2 * n + 1, -1)
    idx = np.concatenate([[0], np.arange(1, 2 * n - 1, 2)])[:n]
    L[idx] = idx * 0 + np.concatenate([[1], np.arange(3, 2 * n + 1, 2)])[:n]; R[idx] = L[idx] + 1
    with pytest.raises(ValueError, match="deeper"):
        boosters.validate_gbt(L, R, np.zeros_like(L), np.zeros(L.size, np.float32), np.zeros(L.size, np.uint8), np.array([0, L.size]), 4)


def test_a_model_document_with_ragged_tree_arrays_is_refused():
    doc = random_model(5, 2, 10, 3)
    t = doc["Model"]["learner"]["gradient_booster"]["model"]["trees"][1]
    t["split_indices"] = t["split_indices"][:-1]
    with pytest.raises(ValueError):
        boosters.XGBTrees.flatten(doc)
    doc = random_model(5, 2, 10, 3)
    doc["Model"]["learner"]["gradient_booster"]["model"]["trees"][0]["split_indices"][0] = 10          # == num_feature
    with pytest.raises(ValueError):
        boosters.XGBTrees.from_raw(ubj(doc), device="cpu")


def test_truncated_or_padded_ubjson_is_refused():
    raw = ubj(random_model(6, 2, 10, 3))
    assert boosters.parse_ubjson(raw)["Model"]["version"] == [2, 0, 3]
    for cut in (1, 7, len(raw) // 3, len(raw) // 2, len(raw) - 1):
        with pytest.raises(ValueError):
            boosters.parse_ubjson(raw[:cut])
    with pytest.raises(ValueError, match="trailing"):
        boosters.parse_ubjson(raw + b"\x00\x00")
    with pytest.raises(ValueError, match="negative"):
        boosters.parse_ubjson(b"[$d#L" + struct.pack(">q", -4))
    # a typed array whose payload is shorter than its count (the remainder a multiple of the item size) is not silently shortened
    with pytest.raises(ValueError, match="truncated"):
        boosters.parse_ubjson(b"[$d#L" + struct.pack(">q", 4) + struct.pack(">2f", 1.0, 2.0))


def test_corrupted_oblivious_trees_are_refused():
    import json
    doc = random_catboost_doc(2, n_trees=3, n_features=8, depth=4)
    good = boosters.CatBoostTrees.flatten(json.loads(json.dumps(doc)))
    boosters.validate_oblivious(*good[:7])
    bad = json.loads(json.dumps(doc)); bad["oblivious_trees"][1]["splits"][0]["float_feature_index"] = -1
    with pytest.raises(ValueError):
        boosters.CatBoostTrees.flatten(bad)
    sf, sb, nt, fs, fl, lv, nf, scale, bias = good
    for args in ((np.where(np.arange(sf.size) == 0, nf, sf), sb, nt, fs, fl, lv, nf),       # feature index == n_features
                 (sf, sb, nt, fs, fl, lv[:-1], nf),                                          # last tree's leaves cut short
                 (sf, sb[:-1], nt, fs, fl, lv, nf),                                          # one border missing
                 (sf, sb, nt[:nf - 1], fs, fl, lv, nf),                                      # NaN flags shorter than the feature count
                 (sf, sb, nt, fs[::-1].copy(), fl, lv, nf)):                                 # offsets not rising
        with pytest.raises(ValueError):
            boosters.validate_oblivious(*args)
        with pytest.raises(ValueError):
            boosters.CatBoostTrees(*args, scale, bias, device="cpu")
