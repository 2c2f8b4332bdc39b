"""GPU: random-forest inference (csrc/forest.hip, trees.py) against scikit-learn's own predict (the oracle for this
third-party arithmetic)."""
import numpy as np
import pytest

from bbbp_amd.trees import ForestGPU

pytestmark = pytest.mark.gpu
ens = pytest.importorskip("sklearn.ensemble")


@pytest.mark.parametrize("cls,kw,n,f", [("RandomForestRegressor", dict(n_estimators=37, max_depth=30, random_state=42), 400, 60),
                                         ("RandomForestRegressor", dict(n_estimators=300, max_depth=6, random_state=1), 300, 20),
                                         ("ExtraTreesRegressor", dict(n_estimators=5, max_depth=None, random_state=3), 250, 33),
                                         ("RandomForestRegressor", dict(n_estimators=1, max_depth=1, random_state=0), 50, 4)])
def test_forest_predict_equals_sklearn(dev, cls, kw, n, f):
    rs = np.random.RandomState(7)
    X = rs.randn(n, f)
    X[:, ::3] = (X[:, ::3] > 0)                 # fingerprint-like binary columns: many ties exactly on thresholds' sides
    y = X @ rs.randn(f) + np.sin(3 * X[:, 1]) + 0.1 * rs.randn(n)
    model = getattr(ens, cls)(**kw).fit(X, y)
    Xt = np.vstack([X[:77], rs.randn(1001, f)])
    got = ForestGPU.from_sklearn(model, device=dev).predict(Xt)
    want = model.predict(Xt)
    # same leaves, float64 sums; scikit-learn adds the trees from several threads, so only the summation order may differ
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_forest_errors_and_empty(dev):
    rs = np.random.RandomState(0)
    X, y = rs.randn(40, 5), rs.randn(40)
    model = ens.RandomForestRegressor(n_estimators=3, random_state=0).fit(X, y)
    f = ForestGPU.from_sklearn(model, device=dev)
    assert f.predict(np.zeros((0, 5))).shape == (0,)
    with pytest.raises(ValueError):
        f.predict(np.zeros((3, 6)))
    with pytest.raises(RuntimeError):
        ForestGPU.from_sklearn(model, device="cpu")


def test_screening_pipeline_matches_columnwise_reference(dev):
    """ensemble.screen = network (eval) + random forest + precomputed column through the linear meta-learner."""
    import torch
    import bbbp_amd
    from bbbp_amd.ensemble import StackedEnsemble, screen
    rs = np.random.RandomState(5)
    n, F = 40, 167
    fp = torch.from_numpy((rs.rand(n, F) > 0.7).astype(np.float32))
    img = torch.from_numpy(rs.rand(n, 49152).astype(np.float32))
    y = rs.randn(n)
    feats = np.hstack([fp.numpy(), img.numpy()])
    rf = ens.RandomForestRegressor(n_estimators=8, max_depth=5, random_state=42).fit(feats, y)
    torch.manual_seed(0)
    model = bbbp_amd.MixedInputModel(F, 128).to(dev)
    xgb_col = rs.randn(n)
    stack = StackedEnsemble.from_coefficients([0.19813994153864287, 0.8730076113813537, 0.16470120078934247], 0.019492486407121146)
    got = screen(model, ForestGPU.from_sklearn(rf, device=dev), stack, fp.to(dev), img.to(dev), extra_columns=(xgb_col,), batch_size=16)
    model.eval()
    with torch.no_grad():
        nn_col = torch.cat([model(fp[i:i + 16].to(dev), img[i:i + 16].to(dev)).reshape(-1) for i in range(0, n, 16)]).double().cpu().numpy()
    want = stack.predict(np.stack([nn_col, rf.predict(feats), xgb_col], axis=1))
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-10, atol=1e-10)



def test_screening_pipeline_with_a_gradient_boosted_column(dev):
    """ensemble.screen with the xgb column computed on the GPU (boosters.XGBTrees) == the same columns assembled by hand with the
    oracle's restatement of XGBoost's predict rule (column order nn, rf, xgb, extra)."""
    import torch
    import bbbp_amd
    from bbbp_amd.boosters import XGBTrees
    from bbbp_amd.ensemble import StackedEnsemble, screen
    from oracle import reference_cpu as oracle
    from test_boosters import random_model, ubj
    rs = np.random.RandomState(6)
    n, F = 24, 167
    fp = torch.from_numpy((rs.rand(n, F) > 0.7).astype(np.float32))
    img = torch.from_numpy(rs.rand(n, 49152).astype(np.float32))
    feats = np.hstack([fp.numpy(), img.numpy()])
    rf = ens.RandomForestRegressor(n_estimators=5, max_depth=4, random_state=1).fit(feats, rs.randn(n))
    bst = XGBTrees.from_raw(ubj(random_model(21, n_trees=30, n_features=F + 49152, depth=8)), device=dev)
    torch.manual_seed(1)
    model = bbbp_amd.MixedInputModel(F, 128).to(dev)
    cat_col = rs.randn(n)
    stack = StackedEnsemble.from_coefficients([0.2, 0.5, 0.2, 0.1], -0.01)
    got = screen(model, ForestGPU.from_sklearn(rf, device=dev), stack, fp.to(dev), img.to(dev), extra_columns=(cat_col,), batch_size=16, boosters=(bst,))
    model.eval()
    with torch.no_grad():
        nn_col = torch.cat([model(fp[i:i + 16].to(dev), img[i:i + 16].to(dev)).reshape(-1) for i in range(0, n, 16)]).double().cpu().numpy()
    a = bst.arrays
    xgb_col = oracle.xgb_predict(a["left"], a["right"], a["feature"], a["cond"], a["default_left"], a["root"], bst.base_score, feats).astype(np.float64)
    want = stack.predict(np.stack([nn_col, rf.predict(feats), xgb_col, cat_col], axis=1))
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-10, atol=1e-10)
