"""GPU prediction for the gradient-boosted base learner of the reference's stacked ensemble.

Reference: ``Models/multi_input_data_regression_opt_transformer_cnn_20250108.py:186-189`` (``XGBRegressor(n_estimators=300,
learning_rate=0.01, max_depth=30, tree_method="hist")`` fitted on ``hstack([fingerprints, images])``) and the fitted model the
repository ships, ``Models/xgb_model_maccs.pkl``.  Fitting stays with XGBoost (third-party CPU/GPU code, as in the reference);
prediction over screening-scale libraries runs here.

The pickle holds the booster's raw model buffer (``Booster.save_raw``: UBJSON).  ``lift_raw_from_pickle`` takes that byte string
out of the pickle stream with ``pickletools`` -- nothing is unpickled, no xgboost import -- and ``parse_ubjson`` /
``XGBTrees.from_raw`` read the documented model schema (``learner.gradient_booster.model.trees[*]``: left_children,
right_children, split_indices, split_conditions, default_left; ``learner.learner_model_param.base_score``).

``CatBoostTrees`` does the same for the cat learner (``CatBoostRegressor(iterations=300, depth=10)``, ...20250108.py:192-195) from
CatBoost's JSON export (``model.save_model(path, format="json")``): oblivious trees over float features.

PARITY UNPINNED: neither xgboost nor catboost is installed in the build image (and the reference ships no fitted CatBoost model), so outputs cannot be compared with the library's own
``predict``; the predict rule is XGBoost's published one (csrc/forest.hip) and the tests check the GPU path against the numpy
restatement in ``oracle/reference_cpu.py`` only.
"""
from __future__ import annotations

import pickletools
import struct
from typing import Any, Tuple

import numpy as np
import torch

from . import _lib, ops

_INT = {"i": (">b", 1), "U": (">B", 1), "I": (">h", 2), "l": (">i", 4), "L": (">q", 8)}
_NUM = dict(_INT, d=(">f", 4), D=(">d", 8))
_NP = {"i": ">i1", "U": ">u1", "I": ">i2", "l": ">i4", "L": ">i8", "d": ">f4", "D": ">f8"}


def parse_ubjson(buf: bytes) -> Any:
    """Universal Binary JSON (ubjson.org, draft 12) -> Python objects; strongly typed arrays come back as numpy arrays."""
    mv = memoryview(buf)
    size = len(mv)

    def need(pos, n):
        if n < 0 or pos + n > size:
            raise ValueError(f"UBJSON: truncated or corrupt document ({n} bytes wanted at byte {pos} of {size})")

    def integer(pos, tag=None):
        if tag is None:
            tag = chr(mv[pos]); pos += 1
        if tag not in _INT:
            raise ValueError(f"UBJSON: integer expected at byte {pos - 1}, found {tag!r}")
        fmt, n = _INT[tag]
        need(pos, n)
        return struct.unpack_from(fmt, mv, pos)[0], pos + n

    def count_of(pos):
        c, pos = integer(pos)
        if c < 0:
            raise ValueError(f"UBJSON: negative container count {c} at byte {pos}")
        return c, pos

    def value(pos, tag=None):
        if tag is None:
            tag = chr(mv[pos]); pos += 1
            while tag == "N":
                tag = chr(mv[pos]); pos += 1
        if tag == "Z":
            return None, pos
        if tag == "T":
            return True, pos
        if tag == "F":
            return False, pos
        if tag in _NUM:
            fmt, n = _NUM[tag]
            need(pos, n)
            return struct.unpack_from(fmt, mv, pos)[0], pos + n
        if tag == "C":
            return chr(mv[pos]), pos + 1
        if tag in ("S", "H"):
            n, pos = count_of(pos)
            need(pos, n)
            return bytes(mv[pos:pos + n]).decode("utf-8"), pos + n
        if tag == "[":
            etype, count = None, None
            if chr(mv[pos]) == "$":
                etype = chr(mv[pos + 1]); pos += 2
                if chr(mv[pos]) != "#":
                    raise ValueError("UBJSON: a typed container needs a count")
            if chr(mv[pos]) == "#":
                count, pos = count_of(pos + 1)
            if etype is not None and etype in _NP:
                n = count * np.dtype(_NP[etype]).itemsize
                need(pos, n)
                return np.frombuffer(mv[pos:pos + n], dtype=_NP[etype]).astype(_NP[etype][1:]), pos + n
            out = []
            if count is not None:
                for _ in range(count):
                    v, pos = value(pos, etype)
                    out.append(v)
                return out, pos
            while chr(mv[pos]) != "]":
                v, pos = value(pos)
                out.append(v)
            return out, pos + 1
        if tag == "{":
            etype, count = None, None
            if chr(mv[pos]) == "$":
                etype = chr(mv[pos + 1]); pos += 2
            if chr(mv[pos]) == "#":
                count, pos = count_of(pos + 1)
            out = {}
            k = 0
            while (count is None and chr(mv[pos]) != "}") or (count is not None and k < count):
                n, pos = count_of(pos)
                need(pos, n)
                key = bytes(mv[pos:pos + n]).decode("utf-8"); pos += n
                out[key], pos = value(pos, etype)
                k += 1
            return out, (pos + 1 if count is None else pos)
        raise ValueError(f"UBJSON: unknown type marker {tag!r} at byte {pos - 1}")

    try:
        obj, end = value(0)
    except (IndexError, struct.error) as e:                    # a marker or number cut off by the end of the buffer
        raise ValueError(f"UBJSON: truncated document ({e})") from None
    if end != size:
        raise ValueError(f"UBJSON: {size - end} trailing bytes after the document")
    return obj


def lift_raw_from_pickle(path: str) -> bytes:
    """The largest bytes / bytearray literal of a pickle stream (an XGBoost estimator's pickle: the booster's raw model buffer),
    read with pickletools.genops: no object of the pickle is constructed."""
    best = b""
    with open(path, "rb") as f:
        for op, arg, _ in pickletools.genops(f.read()):
            if isinstance(arg, (bytes, bytearray)) and len(arg) > len(best):
                best = bytes(arg)
    if not best:
        raise ValueError(f"{path}: no byte string in the pickle stream")
    return best


GBT_MAX_DEPTH = 1024          # csrc/forest.hip: the device walk gives up (NaN) after this many levels


def validate_gbt(left, right, feature, cond, default_left, root, n_features: int) -> int:
    """Host-side check of flattened gbtree arrays BEFORE they reach the GPU (a shipped model file is untrusted input and the device
    walk follows child indices and reads ``x[feature[node]]``): equal array lengths; every node a leaf (both children -1) or an
    internal node whose children lie inside its own tree and whose split index is a feature; no node is the child of two nodes and
    no root is anybody's child (=> every walk from a root ends in a leaf); depth <= GBT_MAX_DEPTH.  Returns the largest depth."""
    left, right, feature, root = (np.asarray(a, np.int64) for a in (left, right, feature, root))
    n = left.size
    if not (right.size == feature.size == np.asarray(cond).size == np.asarray(default_left).size == n):
        raise ValueError("gbtree arrays differ in length")
    if root.ndim != 1 or root.size < 2 or root[0] != 0 or root[-1] != n or np.any(np.diff(root) < 1):
        raise ValueError("tree offsets must rise from 0 to the node count, one node per tree at least")
    if int(n_features) < 1:
        raise ValueError("num_feature must be positive")
    leaf = left < 0
    if np.any(leaf != (right < 0)) or np.any(left[leaf] != -1) or np.any(right[leaf] != -1):
        raise ValueError("a node must have two children or none (-1, -1)")
    tree_of = np.searchsorted(root, np.arange(n), side="right") - 1
    lo, hi = root[tree_of], root[tree_of + 1]
    inner = ~leaf
    for child in (left, right):
        if np.any((child[inner] < lo[inner]) | (child[inner] >= hi[inner])):
            raise ValueError("a child index points outside its tree")
    if np.any((feature[inner] < 0) | (feature[inner] >= int(n_features))):
        raise ValueError(f"a split index is outside [0, {int(n_features)})")
    children = np.concatenate([left[inner], right[inner]])
    if np.unique(children).size != children.size or np.isin(root[:-1], children).any():
        raise ValueError("the nodes do not form trees (a node with two parents, or a root that is a child)")
    frontier, depth = root[:-1], 0
    while True:
        frontier = frontier[inner[frontier]]
        if frontier.size == 0:
            return depth
        depth += 1
        if depth > GBT_MAX_DEPTH:
            raise ValueError(f"tree deeper than {GBT_MAX_DEPTH} levels")
        frontier = np.concatenate([left[frontier], right[frontier]])


class XGBTrees:
    """Flattened regression trees of a gbtree booster + GPU ``predict``."""

    def __init__(self, left, right, feature, cond, default_left, root, n_features: int, base_score: float, device="cuda"):
        self.max_depth = validate_gbt(left, right, feature, cond, default_left, root, n_features)
        self.arrays = dict(left=np.ascontiguousarray(left, np.int32), right=np.ascontiguousarray(right, np.int32),
                           feature=np.ascontiguousarray(feature, np.int32), cond=np.ascontiguousarray(cond, np.float32),
                           default_left=np.ascontiguousarray(default_left, np.uint8), root=np.ascontiguousarray(root, np.int32))
        self.n_trees, self.n_features, self.base_score = len(root) - 1, int(n_features), float(np.float32(base_score))
        self.device = torch.device(device) if device is not None else None
        self._dev = None

    # ---- construction -------------------------------------------------------------------------------------------------
    @staticmethod
    def flatten(model: dict) -> Tuple:
        """(left, right, feature, cond, default_left, root, n_features, base_score) from a parsed XGBoost model document
        (JSON / UBJSON schema of XGBoost >= 1.0; a top-level {"Config", "Model"} pair as written into pickles is accepted)."""
        if "learner" not in model and "Model" in model:
            model = model["Model"]
        learner = model["learner"]
        gb = learner["gradient_booster"]
        if gb.get("name", "gbtree") not in ("gbtree",):
            raise ValueError(f"booster {gb.get('name')!r} is not supported (gbtree only)")
        objective = learner.get("objective", {}).get("name", "reg:squarederror")
        if objective not in ("reg:squarederror", "reg:linear"):
            raise ValueError(f"objective {objective!r} is not supported (the reference fits reg:squarederror)")
        lmp = learner["learner_model_param"]
        if int(lmp.get("num_class", "0")) > 1 or int(lmp.get("num_target", "1")) > 1:
            raise ValueError("single-output regression only")
        base = lmp["base_score"]
        base = float(base.strip("[]")) if isinstance(base, str) else float(np.asarray(base).reshape(-1)[0])
        left, right, feature, cond, dleft, root = [], [], [], [], [], [0]
        for tree in gb["model"]["trees"]:
            cl = np.asarray(tree["left_children"], np.int64); cr = np.asarray(tree["right_children"], np.int64)
            if not (len(cl) == len(cr) == len(tree["split_indices"]) == len(tree["split_conditions"]) == len(tree["default_left"])) or len(cl) == 0:
                raise ValueError("a tree's node arrays differ in length (or are empty)")
            if "split_type" in tree and np.any(np.asarray(tree["split_type"]) != 0):
                raise ValueError("categorical splits are not supported")
            off = root[-1]
            left.append(np.where(cl >= 0, cl + off, -1)); right.append(np.where(cr >= 0, cr + off, -1))
            feature.append(np.asarray(tree["split_indices"], np.int64)); cond.append(np.asarray(tree["split_conditions"], np.float32))
            dleft.append(np.asarray(tree["default_left"], np.uint8))
            root.append(off + len(cl))
        if root[-1] >= 2 ** 31:
            raise ValueError("model too large for 32-bit node indices")
        if len(root) < 2:
            raise ValueError("the model holds no trees")
        flat = (np.concatenate(left), np.concatenate(right), np.concatenate(feature), np.concatenate(cond), np.concatenate(dleft),
                np.asarray(root), int(lmp["num_feature"]))
        validate_gbt(*flat)
        return (*flat, base)

    @classmethod
    def from_raw(cls, raw: bytes, device="cuda") -> "XGBTrees":
        """From ``Booster.save_raw("ubj")`` bytes (what an XGBoost estimator's pickle holds) or a JSON model document."""
        if raw[:1] == b"{" and raw[1:2] in (b'"', b" ", b"\n"):
            import json
            doc = json.loads(raw.decode("utf-8"))
        else:
            doc = parse_ubjson(raw)
        return cls(*cls.flatten(doc), device=device)

    @classmethod
    def from_pickle(cls, path: str, device="cuda") -> "XGBTrees":
        return cls.from_raw(lift_raw_from_pickle(path), device=device)

    # ---- prediction ---------------------------------------------------------------------------------------------------
    def _on_device(self):
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("XGBTrees.predict needs a GPU (no CPU fallback; oracle/reference_cpu.py holds the checker)")
        if self._dev is None:
            self._dev = {k: torch.from_numpy(v).to(self.device) for k, v in self.arrays.items()}
        return self._dev

    def predict_device(self, X, rows_per_call: int = 1 << 16) -> torch.Tensor:
        """``X``: [n, n_features] float32 (numpy or CUDA tensor; NaN = missing).  Returns float32 predictions on the GPU."""
        d = self._on_device()
        Xd = X.to(self.device, torch.float32).contiguous() if isinstance(X, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(self.device)
        if Xd.dim() != 2 or Xd.shape[1] != self.n_features:
            raise ValueError(f"X must be [n, {self.n_features}]")
        n = Xd.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        L = _lib.lib()
        scratch = torch.empty(self.n_trees * min(max(n, 1), rows_per_call), dtype=torch.float32, device=self.device)
        for lo in range(0, n, rows_per_call):
            m = min(rows_per_call, n - lo)
            _lib.check(L.bbbp_gbt_predict(ops._stream(), Xd[lo:lo + m].data_ptr(), m, self.n_features, d["left"].data_ptr(), d["right"].data_ptr(),
                                          d["feature"].data_ptr(), d["cond"].data_ptr(), d["default_left"].data_ptr(), d["root"].data_ptr(),
                                          self.n_trees, self.base_score, scratch.data_ptr(), out[lo:lo + m].data_ptr()), "bbbp_gbt_predict")
        return out

    def predict(self, X) -> np.ndarray:
        return self.predict_device(X).cpu().numpy()


def validate_oblivious(split_feature, split_border, nan_true, tree_first_split, tree_first_leaf, leaf_values, n_features: int) -> None:
    """Host-side check of flattened oblivious trees before they reach the GPU: feature indices inside [0, n_features) (a NaN-treatment
    flag per feature), at most 31 levels per tree, 2^depth leaf values per tree inside ``leaf_values``."""
    sf, fs, fl = np.asarray(split_feature, np.int64), np.asarray(tree_first_split, np.int64), np.asarray(tree_first_leaf, np.int64)
    if int(n_features) < 1 or np.asarray(nan_true).size < int(n_features):
        raise ValueError("n_features must be positive, with one NaN-treatment flag per feature")
    if np.asarray(split_border).size != sf.size:
        raise ValueError("one border per split")
    if sf.size and (sf.min() < 0 or sf.max() >= int(n_features)):
        raise ValueError(f"a float_feature_index is outside [0, {int(n_features)})")
    if fs.ndim != 1 or fs.size < 2 or fs[0] != 0 or fs[-1] != sf.size or np.any(np.diff(fs) < 0) or np.any(np.diff(fs) > 31):
        raise ValueError("split offsets must rise from 0 to the split count, at most 31 levels per tree")
    if fl.size != fs.size - 1 or np.any(fl < 0) or np.any(fl + (1 << np.diff(fs)) > np.asarray(leaf_values).size):
        raise ValueError("every tree needs 2^depth leaf values inside leaf_values")


class CatBoostTrees:
    """Oblivious trees of a CatBoost model exported as JSON + GPU ``predict`` (float features only, single-dimensional output)."""

    def __init__(self, split_feature, split_border, nan_true, tree_first_split, tree_first_leaf, leaf_values, n_features: int,
                 scale: float = 1.0, bias: float = 0.0, device="cuda"):
        validate_oblivious(split_feature, split_border, nan_true, tree_first_split, tree_first_leaf, leaf_values, n_features)
        self.arrays = dict(split_feature=np.ascontiguousarray(split_feature, np.int32), split_border=np.ascontiguousarray(split_border, np.float32),
                           nan_true=np.ascontiguousarray(nan_true, np.uint8), tree_first_split=np.ascontiguousarray(tree_first_split, np.int32),
                           tree_first_leaf=np.ascontiguousarray(tree_first_leaf, np.int64), leaf_values=np.ascontiguousarray(leaf_values, np.float64))
        self.n_trees, self.n_features, self.scale, self.bias = len(tree_first_split) - 1, int(n_features), float(scale), float(bias)
        self.device = torch.device(device) if device is not None else None
        self._dev = None

    @staticmethod
    def flatten(doc: dict, n_features: int | None = None) -> Tuple:
        """(split_feature, split_border, nan_true, tree_first_split, tree_first_leaf, leaf_values, n_features, scale, bias) from the
        JSON export's "oblivious_trees" / "features_info" / "scale_and_bias".  Feature indices are positions in the flat feature
        vector (``flat_feature_index``)."""
        feats = doc.get("features_info", {})
        if feats.get("categorical_features") or feats.get("text_features") or feats.get("embedding_features"):
            raise ValueError("float features only")
        ff = feats.get("float_features", [])
        flat = {int(f.get("feature_index", i)): int(f.get("flat_feature_index", f.get("feature_index", i))) for i, f in enumerate(ff)}
        nf = n_features if n_features is not None else (max(flat.values()) + 1 if flat else 0)
        nan_true = np.zeros(max(nf, 1), np.uint8)
        for i, f in enumerate(ff):
            if f.get("nan_value_treatment", "AsIs") == "AsTrue":
                nan_true[flat[int(f.get("feature_index", i))]] = 1
        sf, sb, first_split, first_leaf, leaves = [], [], [0], [0], []
        for tree in doc["oblivious_trees"]:
            splits = tree.get("splits", [])
            if len(splits) > 31:
                raise ValueError("trees deeper than 31 levels are not supported")
            for sp in splits:
                if sp.get("split_type", "FloatFeature") != "FloatFeature":
                    raise ValueError("float-feature splits only")
                fi = int(sp["float_feature_index"])
                if fi < 0:
                    raise ValueError("negative float_feature_index")
                sf.append(flat.get(fi, fi)); sb.append(float(sp["border"]))
            lv = np.asarray(tree["leaf_values"], np.float64)
            if lv.size != 1 << len(splits):
                raise ValueError("single-dimensional leaf values expected (2^depth per tree)")
            leaves.append(lv)
            first_split.append(len(sf)); first_leaf.append(first_leaf[-1] + lv.size)
        sab = doc.get("scale_and_bias", [1.0, [0.0]])
        bias = sab[1][0] if isinstance(sab[1], (list, tuple)) else sab[1]
        if sf and max(sf) >= nf:
            nf = max(sf) + 1
            nan_true = np.concatenate([nan_true, np.zeros(nf - len(nan_true), np.uint8)])
        return (np.asarray(sf, np.int32), np.asarray(sb, np.float32), nan_true[:max(nf, 1)], np.asarray(first_split, np.int32),
                np.asarray(first_leaf[:-1], np.int64), np.concatenate(leaves) if leaves else np.zeros(0), nf,
                float(sab[0]), float(bias))

    @classmethod
    def from_json(cls, text_or_doc, n_features: int | None = None, device="cuda") -> "CatBoostTrees":
        import json
        doc = json.loads(text_or_doc) if isinstance(text_or_doc, (str, bytes)) else text_or_doc
        return cls(*cls.flatten(doc, n_features), device=device)

    def predict_device(self, X, rows_per_call: int = 1 << 16) -> torch.Tensor:
        """``X``: [n, n_features] float32 (numpy or CUDA tensor; NaN = missing).  Returns float64 predictions on the GPU."""
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("CatBoostTrees.predict needs a GPU (no CPU fallback; oracle/reference_cpu.py holds the checker)")
        if self._dev is None:
            self._dev = {k: torch.from_numpy(v).to(self.device) for k, v in self.arrays.items()}
        d = self._dev
        Xd = X.to(self.device, torch.float32).contiguous() if isinstance(X, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(self.device)
        if Xd.dim() != 2 or Xd.shape[1] != self.n_features:
            raise ValueError(f"X must be [n, {self.n_features}]")
        n = Xd.shape[0]
        out = torch.empty(n, dtype=torch.float64, device=self.device)
        scratch = torch.empty(self.n_trees * min(max(n, 1), rows_per_call), dtype=torch.float64, device=self.device)
        L = _lib.lib()
        for lo in range(0, n, rows_per_call):
            m = min(rows_per_call, n - lo)
            _lib.check(L.bbbp_oblivious_predict(ops._stream(), Xd[lo:lo + m].data_ptr(), m, self.n_features, d["split_feature"].data_ptr(),
                                                d["split_border"].data_ptr(), d["nan_true"].data_ptr(), d["tree_first_split"].data_ptr(),
                                                d["tree_first_leaf"].data_ptr(), d["leaf_values"].data_ptr(), self.n_trees, self.scale, self.bias,
                                                scratch.data_ptr(), out[lo:lo + m].data_ptr()), "bbbp_oblivious_predict")
        return out

    def predict(self, X) -> np.ndarray:
        return self.predict_device(X).cpu().numpy()
