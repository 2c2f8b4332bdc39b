"""CPU: stacked-ensemble surface against the oracle, scikit-learn and the reference's shipped meta-learners."""
import numpy as np
import pytest
import torch

from bbbp_amd.ensemble import StackedEnsemble, weighted_ensemble
from oracle import reference_cpu as oracle


def test_known_answer_meta_learners():
    rng = np.random.default_rng(1)
    X = rng.normal(size=(50, 3))
    for name, (coef, icpt) in oracle.STACKED_KNOWN.items():
        m = StackedEnsemble.from_coefficients(coef, icpt, alpha=1.0 if "maccs_opt" in name else 0.0)
        np.testing.assert_allclose(m.predict(X), oracle.linear_predict(X, coef, icpt), rtol=0, atol=1e-15)
        np.testing.assert_allclose(m.predict_device(torch.from_numpy(X[:, 0]), X[:, 1], X[:, 2]).numpy(), m.predict(X), atol=1e-14)
    # single-molecule call shape used by the reference: predict([[nn, rf, xgb]])  (..._opt.py:202-203)
    m = StackedEnsemble.from_coefficients(*oracle.STACKED_KNOWN["stacked_model.pkl"])
    assert m.predict([[0.1, 0.2, 0.3]]).shape == (1,)


@pytest.mark.parametrize("alpha", [0.0, 1.0])
def test_fit_matches_sklearn_and_oracle(alpha):
    from sklearn.linear_model import LinearRegression, Ridge
    rng = np.random.default_rng(0)
    X = rng.normal(size=(1058, 4))                       # [nn, rf, xgb, cat] out-of-fold columns, B3DB size
    y = X @ np.array([0.2, 0.5, 0.2, 0.1]) + 0.02 + 0.1 * rng.normal(size=1058)
    sk = (Ridge(alpha=1.0) if alpha else LinearRegression()).fit(X, y)
    m = StackedEnsemble(alpha).fit(X, y)
    np.testing.assert_allclose(m.coef_, sk.coef_, rtol=1e-9)
    np.testing.assert_allclose(m.intercept_, sk.intercept_, rtol=1e-9)
    np.testing.assert_allclose(m.predict(X), sk.predict(X), rtol=1e-10)
    c, b = oracle.linear_fit(X, y, alpha)
    np.testing.assert_allclose(m.coef_, c, rtol=1e-12)
    assert abs(m.intercept_ - b) < 1e-12


def test_weighted_and_errors():
    np.testing.assert_allclose(weighted_ensemble([1, 2], [3, 4], [5, 6]), oracle.weighted_ensemble([1, 2], [3, 4], [5, 6]))
    with pytest.raises(RuntimeError):
        StackedEnsemble().predict([[1, 2, 3]])
    with pytest.raises(ValueError):
        StackedEnsemble.from_coefficients([1, 2, 3], 0.0).predict([[1, 2]])
    with pytest.raises(ValueError):
        StackedEnsemble().fit(np.zeros((3, 2)), np.zeros(4))


def test_forest_flattening_walks_to_sklearn_predictions():
    """Host side of trees.ForestGPU (no GPU): the concatenated node arrays, walked in numpy with scikit-learn's rule
    (float32 feature <= float64 threshold goes left), give rf.predict."""
    ens = pytest.importorskip("sklearn.ensemble")
    from bbbp_amd.trees import ForestGPU
    rs = np.random.RandomState(3)
    X = rs.randn(120, 9); X[:, ::2] = X[:, ::2] > 0
    y = X @ rs.randn(9) + 0.1 * rs.randn(120)
    rf = ens.RandomForestRegressor(n_estimators=7, max_depth=6, random_state=42).fit(X, y)
    left, right, feature, threshold, value, root, nf = ForestGPU.flatten_sklearn(rf)
    assert nf == 9 and len(root) == 8 and root[0] == 0 and root[-1] == len(left) == len(value)
    Xt = rs.randn(50, 9).astype(np.float32)
    got = np.zeros(50)
    for i in range(50):
        for t in range(7):
            node = root[t]
            while left[node] >= 0:
                node = left[node] if np.float64(Xt[i, feature[node]]) <= threshold[node] else right[node]
            got[i] += value[node]
    np.testing.assert_allclose(got / 7, rf.predict(Xt), rtol=1e-12, atol=1e-12)


def test_stacking_regressor_semantics_match_sklearn():
    """The published script's stack (…20250113.py:394-403) is sklearn's StackingRegressor fitted ON the [N,4] out-of-fold matrix:
    base learners refit on those four columns, the final LinearRegression sees their 5-fold cross_val_predict.  Same numbers as
    scikit-learn's own class with scikit-learn base learners (RandomForest / ExtraTrees stand in for the absent boosters)."""
    from sklearn.ensemble import ExtraTreesRegressor, RandomForestRegressor
    from sklearn.ensemble import StackingRegressor as SkStack
    from sklearn.linear_model import LinearRegression
    from bbbp_amd.ensemble import StackingRegressor
    rng = np.random.default_rng(7)
    X = rng.normal(size=(203, 4))                         # 203: KFold(5) folds of unequal size
    y = X @ np.array([0.2, 0.5, 0.2, 0.1]) + 0.1 * np.sin(3 * X[:, 0]) + 0.05 * rng.normal(size=203)
    ests = [("rf", RandomForestRegressor(n_estimators=20, max_depth=8, random_state=42)),
            ("et", ExtraTreesRegressor(n_estimators=10, max_depth=6, random_state=42))]
    sk = SkStack(estimators=ests, final_estimator=LinearRegression()).fit(X, y)
    mine = StackingRegressor(ests).fit(X, y)
    np.testing.assert_allclose(mine.final_estimator_.coef_, sk.final_estimator_.coef_, rtol=1e-9)
    np.testing.assert_allclose(mine.final_estimator_.intercept_, sk.final_estimator_.intercept_, rtol=1e-9, atol=1e-12)
    Xt = rng.normal(size=(50, 4))
    np.testing.assert_allclose(mine.predict(Xt), sk.predict(Xt), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(mine.transform(Xt), sk.transform(Xt), rtol=1e-12, atol=1e-12)
    assert not np.allclose(mine.predict(Xt), StackedEnsemble().fit(X, y).predict(Xt), atol=1e-3)   # it is NOT X c + b on the columns
    with pytest.raises(ValueError):
        StackingRegressor(ests).fit(X[:3], y[:3])
