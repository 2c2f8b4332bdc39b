"""GPU: the batched MLP-classifier trainer (csrc/mlp.hip, mlp.py) against scikit-learn's MLPClassifier with the same
hyper-parameters and random_state (third-party arithmetic, scikit-learn is the oracle)."""
import warnings

import numpy as np
import pytest

from bbbp_amd.mlp import GridMLPTrainer, MLPConfig, grid_search_cv

pytestmark = pytest.mark.gpu
sk = pytest.importorskip("sklearn.neural_network")


def make_data(n, f, seed):
    rs = np.random.RandomState(seed)
    X = rs.randn(n, f)
    w = rs.randn(f)
    y = ((X @ w + 0.5 * rs.randn(n)) > 0).astype(np.float64)
    return X, y


def sk_fit(X, y, cfg):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = sk.MLPClassifier(hidden_layer_sizes=tuple(cfg.hidden_layer_sizes), activation=cfg.activation, solver="adam", alpha=cfg.alpha,
                             batch_size=cfg.batch_size, learning_rate_init=cfg.learning_rate_init, max_iter=cfg.max_iter, tol=cfg.tol,
                             n_iter_no_change=cfg.n_iter_no_change, random_state=cfg.random_state, shuffle=True)
        rows = np.arange(len(y)) if cfg.train_rows is None else cfg.train_rows
        m.fit(X[rows], y[rows])
    return m


@pytest.mark.parametrize("hidden,act,bs,lr", [((100,), "relu", 32, 0.001), ((100, 50), "tanh", 64, 0.01), ((200, 100), "relu", 128, 0.001),
                                              ((20,), "tanh", 7, 0.1)])
def test_few_epochs_follow_sklearn_step_for_step(dev, hidden, act, bs, lr):
    X, y = make_data(333, 100, 1)          # 333 rows: a ragged last mini-batch
    cfg = MLPConfig(hidden_layer_sizes=hidden, activation=act, batch_size=bs, learning_rate_init=lr, max_iter=4, random_state=3)
    ref = sk_fit(X, y, cfg)
    got = GridMLPTrainer(X, y, device=dev).fit([cfg], epochs_per_launch=3)[0]
    assert got.n_iter_ == ref.n_iter_ == 4
    np.testing.assert_allclose(got.loss_curve_, ref.loss_curve_, rtol=1e-9, atol=1e-12)
    # summation order inside the products differs from BLAS: last-bit differences, amplified by lr = 0.1 / batch 7
    tol = 1e-4 if lr >= 0.1 else 1e-6      # (scikit-learn's own BLAS threading changes its last bits from host to host)
    for a, b in zip(got.coefs_ + got.intercepts_, ref.coefs_ + ref.intercepts_):
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol * 1e-2)
    Xt, _ = make_data(50, 100, 2)
    np.testing.assert_allclose(got.predict_proba(Xt), ref.predict_proba(Xt), rtol=tol, atol=tol * 1e-2)
    assert (got.predict(Xt) == ref.predict(Xt)).all()


def test_stopping_rule_and_many_models_at_once(dev):
    """Training-loss stopping (tol / n_iter_no_change) ends each fit at scikit-learn's epoch; several fits with different
    folds, sizes and seeds run in ONE launch sequence and do not disturb each other."""
    X, y = make_data(400, 100, 5)
    rs = np.random.RandomState(0)
    cfgs = []
    for i, (hidden, act, lr, bs) in enumerate([((100,), "relu", 0.01, 32), ((100, 50), "relu", 0.01, 64), ((100,), "tanh", 0.1, 128),
                                               ((200, 100), "tanh", 0.01, 64), ((100,), "relu", 0.1, 32)]):
        rows = np.sort(rs.choice(400, 320, replace=False))
        cfgs.append(MLPConfig(hidden_layer_sizes=hidden, activation=act, learning_rate_init=lr, batch_size=bs, max_iter=60,
                              random_state=10 + i, train_rows=rows))
    fitted = GridMLPTrainer(X, y, device=dev).fit(cfgs, epochs_per_launch=8)
    for cfg, got in zip(cfgs, fitted):
        ref = sk_fit(X, y, cfg)
        # chaotic amplification of last-bit differences over thousands of Adam steps: compare the trajectory loosely
        assert abs(got.n_iter_ - ref.n_iter_) <= 3, (got.n_iter_, ref.n_iter_)
        k = min(got.n_iter_, ref.n_iter_, 10)
        np.testing.assert_allclose(got.loss_curve_[:k], ref.loss_curve_[:k], rtol=1e-6)
        agree = (got.predict(X) == ref.predict(X)).mean()
        assert agree >= 0.97, agree


def test_grid_search_cv_matches_sklearn_choice(dev):
    X, y = make_data(300, 100, 9)
    grid = {"hidden_layer_sizes": [(100,), (100, 50)], "activation": ["relu", "tanh"], "learning_rate_init": [0.001, 0.1], "batch_size": [64]}
    base = MLPConfig(max_iter=15)
    best, scores, fitted = grid_search_cv(X, y, grid, cv=3, base=base, device=dev, random_state=0)
    assert len(scores) == 8 and len(fitted) == 24 and all(0.0 <= s <= 1.0 for s in scores)
    from sklearn.model_selection import GridSearchCV
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gs = GridSearchCV(sk.MLPClassifier(max_iter=15, random_state=0), {**grid, "solver": ["adam"]}, cv=3, scoring="f1").fit(X, y)
    keys = sorted(grid)
    ref_scores = {tuple(p[k] for k in keys): s for p, s in zip(gs.cv_results_["params"], gs.cv_results_["mean_test_score"])}
    from itertools import product
    ours = {vals: s for vals, s in zip(product(*(grid[k] for k in keys)), scores)}
    for kk in ours:
        assert abs(ours[kk] - ref_scores[kk]) <= 0.03, (kk, ours[kk], ref_scores[kk])


def test_mlp_errors(dev):
    X, y = make_data(20, 5, 0)
    with pytest.raises(RuntimeError):
        GridMLPTrainer(X, y, device="cpu")
    with pytest.raises(ValueError):
        GridMLPTrainer(X, y + 2, device=dev)
    with pytest.raises(ValueError):
        GridMLPTrainer(X, y, device=dev).fit([MLPConfig(activation="logistic")])
    assert GridMLPTrainer(X, y, device=dev).fit([]) == []


def test_fits_that_share_a_random_stream_equal_the_same_fits_run_alone(dev):
    """Round 4: fits with the same (integer seed, layer sizes, number of training rows) consume ONE RandomState stream on the host (initial
    weights and shuffles drawn once per group).  Five such fits -- other folds, batch sizes, learning rates, epoch counts -- in one call
    must give, bit for bit, what each gives when it is the only fit of a call (where it owns its stream)."""
    X, y = make_data(400, 100, 9)
    rs = np.random.RandomState(1)
    cfgs = []
    for i, (bs, lr, epochs) in enumerate([(32, 0.01, 5), (64, 0.01, 3), (128, 0.1, 7), (32, 0.001, 2), (64, 0.1, 6)]):
        rows = np.sort(rs.choice(400, 320, replace=False))
        cfgs.append(MLPConfig(hidden_layer_sizes=(100, 50), activation="relu", learning_rate_init=lr, batch_size=bs, max_iter=epochs, tol=0.0,
                              n_iter_no_change=10 ** 9, random_state=0, train_rows=rows))
    together = GridMLPTrainer(X, y, device=dev).fit(cfgs, epochs_per_launch=4)
    for cfg, got in zip(cfgs, together):
        alone = GridMLPTrainer(X, y, device=dev).fit([cfg], epochs_per_launch=4)[0]
        assert got.n_iter_ == alone.n_iter_ == cfg.max_iter
        assert got.loss_curve_ == alone.loss_curve_
        for a, b in zip(got.coefs_ + got.intercepts_, alone.coefs_ + alone.intercepts_):
            assert np.array_equal(a, b)
