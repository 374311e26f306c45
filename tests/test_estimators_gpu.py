"""GPU tests of the estimator surface: the reference's own regression files and API
behaviour tests (tests/test_regressions.py, tests/test_estimators.py), re-written against
``sknnr_amd`` -- same inputs, same expected values."""

from __future__ import annotations

import pickle

import numpy as np
import pytest
from sklearn.exceptions import NotFittedError

from conftest import assert_neighbors_match, load_golden, yaimpute_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import sknnr_amd
    from sknnr_amd import _native

    assert _native.device_count() >= 1
    return sknnr_amd


def _estimators(E):
    return {
        "raw": (E.RawKNNRegressor, {}), "euclidean": (E.EuclideanKNNRegressor, {}),
        "mahalanobis": (E.MahalanobisKNNRegressor, {}), "gnn": (E.GNNRegressor, {}),
        "msn": (E.MSNRegressor, {}),
    }


CASES = [("raw", None), ("euclidean", None), ("mahalanobis", None), ("gnn", None), ("gnn", 3), ("msn", None), ("msn", 3)]


@pytest.mark.parametrize(("name", "n_components"), CASES)
@pytest.mark.parametrize("reference", [True, False], ids=["reference", "target"])
def test_reference_regression_files_kneighbors(E, moscow_frames, name, n_components, reference):
    """REF tests/test_regressions.py:57-86 against the reference's own .npz data."""
    cls, kw = _estimators(E)[name]
    if n_components is not None:
        kw = dict(kw, n_components=n_components)
    est = cls(n_neighbors=5, **kw).fit(moscow_frames["X_train"], moscow_frames["y_train"])
    size = "reduced" if n_components else "full"
    which = "reference" if reference else "target"
    X = None if reference else moscow_frames["X_test"]
    for ids in (True, False):
        dist, nn = est.kneighbors(X, return_dataframe_index=ids)
        r = load_golden(f"ref_regressions/test_kneighbors_{which}_{size}_{name}_k5_{'ids' if ids else 'index'}_.npz")
        assert nn.dtype == r["nn"].dtype
        assert_neighbors_match(nn, dist, r["nn"], r["dist"])


@pytest.mark.parametrize(("name", "n_components"), CASES)
@pytest.mark.parametrize("weighted", [True, False], ids=["weighted", "unweighted"])
def test_reference_regression_files_predict(E, moscow_frames, name, n_components, weighted):
    """REF tests/test_regressions.py:89-122."""
    cls, kw = _estimators(E)[name]
    if n_components is not None:
        kw = dict(kw, n_components=n_components)
    weights = yaimpute_weights if weighted else "uniform"
    est = cls(n_neighbors=5, weights=weights, **kw).fit(moscow_frames["X_train"], moscow_frames["y_train"])
    size = "reduced" if n_components else "full"
    wname = "weighted" if weighted else "unweighted"
    r = load_golden(f"ref_regressions/test_predict_reference_{wname}_{size}_{name}_k5_.npz")
    np.testing.assert_allclose(est.independent_prediction_, r["pred"], rtol=1e-5, atol=1e-8)
    assert est.independent_score_ == pytest.approx(float(r["score"]), rel=1e-5)
    r = load_golden(f"ref_regressions/test_predict_target_{wname}_{size}_{name}_k5_.npz")
    np.testing.assert_allclose(est.predict(moscow_frames["X_test"]), r["pred"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("name", ["raw", "euclidean", "mahalanobis", "gnn_full", "gnn_reduced", "msn_full", "msn_reduced"])
def test_extended_moscow_goldens(E, moscow_frames, name):
    """Vectors generated from the reference (tests/golden/make_golden.py): k in {1,5,7},
    'distance' weights, deterministic ordering off, fitted transform matrices."""
    g = load_golden(f"moscow_{name}.npz")
    base = name.split("_")[0]
    cls, kw = _estimators(E)[base]
    if name.endswith("reduced"):
        kw = dict(kw, n_components=3)
    Xtr, ytr, Xte = moscow_frames["X_train"], moscow_frames["y_train"], moscow_frames["X_test"]
    for k in (1, 5, 7):
        est = cls(n_neighbors=k, **kw).fit(Xtr, ytr)
        reg = getattr(est, "regressor_", est)
        if k == 5:
            assert reg._fit_method == str(g["fit_method"])
            assert est.n_features_in_ == int(g["n_features_in_"])
            if hasattr(est, "transformer_"):
                for attr in ("projector_", "env_center_", "transform_"):
                    if "tr_" + attr in g:
                        np.testing.assert_allclose(getattr(est.transformer_, attr), g["tr_" + attr], rtol=1e-8, atol=1e-10)
                np.testing.assert_allclose(reg._fit_X, g["Xt_train"], rtol=1e-9, atol=1e-10)
        d, i = est.kneighbors()
        assert_neighbors_match(i, d, g[f"kn_ref_k{k}_nn"], g[f"kn_ref_k{k}_dist"])
        d, i = est.kneighbors(Xte)
        assert_neighbors_match(i, d, g[f"kn_tgt_k{k}_nn"], g[f"kn_tgt_k{k}_dist"])
        np.testing.assert_array_equal(est.kneighbors(Xte, return_distance=False, return_dataframe_index=True),
                                      g[f"kn_tgt_k{k}_ids"])
        if k == 5:
            d, i = est.kneighbors(Xte, use_deterministic_ordering=False)
            assert_neighbors_match(i, d, g["kn_tgt_k5_nd_nn"], g["kn_tgt_k5_nd_dist"])
            np.testing.assert_allclose(est.predict(Xte), g["pred_tgt_uniform"], rtol=1e-5, atol=1e-8)
            assert est.score(Xte, moscow_frames["y_test"]) == pytest.approx(float(g["score_tgt_uniform"]), rel=1e-5)
    for wname, w in (("distance", "distance"), ("yaimpute", yaimpute_weights)):
        for k in (5, 7):
            est = cls(n_neighbors=k, weights=w, **kw).fit(Xtr, ytr)
            np.testing.assert_allclose(est.independent_prediction_, g[f"indep_pred_{wname}_k{k}"], rtol=1e-5, atol=1e-8)
            assert est.independent_score_ == pytest.approx(float(g[f"indep_score_{wname}_k{k}"]), rel=1e-5)
            np.testing.assert_allclose(est.predict(Xte), g[f"pred_tgt_{wname}_k{k}"], rtol=1e-5, atol=1e-8)


def test_baseline_config_1_msn_on_swo(E):
    """BASELINE.json configs[0]: MSNRegressor on load_swo_ecoplot(), k=5."""
    from sknnr_amd.datasets import load_swo_ecoplot

    g = load_golden("swo_msn_k5.npz")
    X, y = load_swo_ecoplot(return_X_y=True, as_frame=True)
    est = E.MSNRegressor(n_neighbors=5).fit(X, y)
    assert est.n_features_in_ == int(g["n_features_in_"]) == 17
    assert est.regressor_._fit_method == str(g["fit_method"]) == "brute"
    assert est.independent_score_ == pytest.approx(float(g["indep_score"]), rel=1e-7)
    assert est.independent_score_ == pytest.approx(0.15023, abs=1e-5)
    np.testing.assert_allclose(est.independent_prediction_[::16], g["indep_pred_rows"], rtol=1e-5, atol=1e-8)
    d, i = est.kneighbors()
    assert_neighbors_match(i, d, g["kn_ref_nn"], g["kn_ref_dist"])
    d, i = est.kneighbors(X)
    assert_neighbors_match(i, d, g["kn_self_nn"], g["kn_self_dist"], atol=1e-6)
    np.testing.assert_allclose(est.predict(X)[::16], g["pred_rows"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("name", ["euclidean", "mahalanobis", "gnn", "msn"])
def test_synthetic_estimator_goldens(E, name):
    """Small versions of BASELINE configs 2-5 on the synthetic law, fitted by the reference."""
    from sknnr_amd import synth

    g = load_golden(f"synth_est_{name}.npz")
    cfg = {
        "euclidean": (E.EuclideanKNNRegressor, dict(n_neighbors=5), "linear"),
        "mahalanobis": (E.MahalanobisKNNRegressor, dict(n_neighbors=5), "linear"),
        "gnn": (E.GNNRegressor, dict(n_neighbors=7, weights="distance"), "positive"),
        "msn": (E.MSNRegressor, dict(n_neighbors=1, n_components=8), "linear"),
    }[name]
    x_ref, y, x_q = synth.make_problem(1500, 512, 16, t=20, kind=cfg[2])
    est = cfg[0](**cfg[1]).fit(x_ref, y)
    assert est.n_features_in_ == int(g["n_features_in_"])
    assert est.regressor_._fit_method == str(g["fit_method"])
    d, i = est.kneighbors(x_q)
    assert_neighbors_match(i, d, g["nn"], g["dist"])
    np.testing.assert_allclose(est.predict(x_q), g["pred"], rtol=1e-5, atol=1e-8)
    assert est.independent_score_ == pytest.approx(float(g["indep_score"]), rel=1e-6)
    np.testing.assert_allclose(est.independent_prediction_[::8], g["indep_pred_rows"], rtol=1e-5, atol=1e-8)


# ---- API behaviour (REF tests/test_estimators.py:156-378) ----------------------------------

ALL = ["raw", "euclidean", "mahalanobis", "gnn", "msn"]


@pytest.mark.parametrize("name", ALL)
def test_not_fitted_raises(E, moscow, name):
    cls, kw = _estimators(E)[name]
    with pytest.raises(NotFittedError):
        cls(**kw).kneighbors(moscow["X_train"])
    with pytest.raises(NotFittedError):
        cls(**kw).predict(moscow["X_train"])


@pytest.mark.parametrize("name", ALL)
def test_dataframe_indexes(E, name):
    """REF tests/test_estimators.py:181-201."""
    from sknnr_amd.datasets import load_moscow_stjoes

    cls, kw = _estimators(E)[name]
    est = cls(n_neighbors=1, **kw)
    ds = load_moscow_stjoes()
    X_df, _ = load_moscow_stjoes(as_frame=True, return_X_y=True)
    est.fit(ds.data, ds.target)
    with pytest.raises(NotFittedError, match="fitted with a dataframe"):
        est.kneighbors(return_dataframe_index=True)
    est.fit(ds.data.tolist(), ds.target)
    assert not hasattr(est, "dataframe_index_in_")
    est.fit(X_df, ds.target)
    np.testing.assert_array_equal(est.dataframe_index_in_, ds.index)
    idx = est.kneighbors(X_df, return_distance=False, return_dataframe_index=True)
    np.testing.assert_array_equal(idx.ravel(), ds.index)


@pytest.mark.parametrize("name", ALL)
def test_lists_dataframes_and_output_types(E, moscow, moscow_frames, name):
    cls, kw = _estimators(E)[name]
    est = cls(**kw).fit(moscow["X_train"].tolist(), moscow["y_train"].tolist())
    p = est.predict(moscow["X_test"].tolist())
    assert isinstance(p, np.ndarray) and p.shape == (33, 35)
    est = cls(**kw).fit(moscow_frames["X_train"], moscow_frames["y_train"])
    assert isinstance(est.predict(moscow_frames["X_test"]), np.ndarray)
    with pytest.warns(UserWarning, match="fitted with feature names"):
        est.predict(moscow["X_test"])
    d, i = est.kneighbors(moscow_frames["X_test"])
    assert d.dtype == np.float64 and i.dtype == np.int64 and d.shape == i.shape == (33, 5)


@pytest.mark.parametrize("name", ["gnn", "msn"])
def test_y_fit(E, moscow, name):
    cls, kw = _estimators(E)[name]
    X, y = moscow["X_train"], moscow["y_train"]
    y_fit = y[:, :10] + 0.25  # strictly positive rows (CCA needs positive row sums)
    est = cls(**kw).fit(X, y)
    assert est.y_fit_ is None
    without = est.independent_prediction_
    est.fit(X, y, y_fit=y_fit)
    np.testing.assert_array_equal(est.y_fit_, y_fit)
    assert not np.array_equal(est.independent_prediction_, without)


@pytest.mark.parametrize("name", ["euclidean", "gnn"])
def test_gridsearch_and_pickle(E, moscow, name):
    from sklearn.model_selection import GridSearchCV

    cls, kw = _estimators(E)[name]
    X, y = moscow["X_train"], moscow["y_train"]
    gs = GridSearchCV(cls(**kw), param_grid={"n_neighbors": [1, 3]}, cv=2)
    gs.fit(X, y)
    gs.predict(X)
    est = cls(**kw).fit(X, y)
    clone = pickle.loads(pickle.dumps(est))
    np.testing.assert_array_equal(clone.predict(moscow["X_test"]), est.predict(moscow["X_test"]))


def test_transformed_feature_count(E, moscow):
    for name in ("euclidean", "mahalanobis", "gnn", "msn"):
        cls, kw = _estimators(E)[name]
        est = cls(**kw).fit(moscow["X_train"], moscow["y_train"])
        assert est.transformer_.n_features_in_ == 28
        assert est.n_features_in_ == len(est.transformer_.get_feature_names_out())


@pytest.mark.parametrize(("deterministic", "expected"), [(False, [1, 0]), (True, [0, 1])])
def test_kneighbors_deterministic_ordering(E, deterministic, expected):
    """REF tests/test_estimators.py:306-327."""
    X = np.array([1e-11, 1e-12, 1.0]).reshape(-1, 1)
    est = E.RawKNNRegressor(n_neighbors=2).fit(X, np.array([0, 1, 2]))
    _, idx = est.kneighbors(np.array([[0.0]]), use_deterministic_ordering=deterministic)
    assert idx[0].tolist() == expected


def test_kneighbors_uses_index_difference(E):
    """REF tests/test_estimators.py:330-348."""
    X = np.array([1e-11, 1e-12, 1.0]).reshape(-1, 1)
    est = E.RawKNNRegressor(n_neighbors=2).fit(X, np.array([0, 1, 2]))
    _, idx = est.kneighbors(np.array([[0.0], [0.0]]))
    assert idx.tolist() == [[0, 1], [1, 0]]


@pytest.mark.parametrize(("decimals", "expected"), [(8, [2, 1, 0]), (5, [1, 2, 0]), (2, [0, 1, 2])])
def test_kneighbors_precision_decimals(E, monkeypatch, decimals, expected):
    """REF tests/test_estimators.py:351-378."""
    monkeypatch.setattr(E.RawKNNRegressor, "DISTANCE_PRECISION_DECIMALS", decimals)
    X = np.array([1e-3, 1e-6, 1e-9, 1.0]).reshape(-1, 1)
    est = E.RawKNNRegressor(n_neighbors=3).fit(X, np.array([0, 1, 2, 3]))
    _, idx = est.kneighbors(np.array([[0.0]]))
    assert idx[0].tolist() == expected


def test_k_too_large_is_a_value_error(E, moscow):
    est = E.RawKNNRegressor(n_neighbors=3).fit(moscow["X_train"][:6], moscow["y_train"][:6])
    with pytest.raises(ValueError, match="Expected n_neighbors <= n_samples_fit"):
        est.kneighbors(moscow["X_test"], n_neighbors=7)
    with pytest.raises(NotImplementedError, match="Euclidean"):
        E.RawKNNRegressor(metric="manhattan").fit(moscow["X_train"], moscow["y_train"])


def test_sharded_single_rank_over_rccl(E, moscow):
    """The N > 1 machinery with one rank on the real backend (nccl = RCCL): contiguous shards and
    the chunk-cyclic in-place gather both reproduce the plain call, raw and transformed."""
    import os
    import socket

    import torch
    import torch.distributed as dist

    from sknnr_amd.distributed import ShardedKNN

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    except Exception as err:  # no usable RCCL in this environment: nothing of ours to test
        pytest.skip(f"RCCL process group could not be created: {err}")
    try:
        for cls in (E.RawKNNRegressor, E.GNNRegressor):
            est = cls(n_neighbors=3).fit(moscow["X_train"], moscow["y_train"])
            x = np.tile(moscow["X_test"], (40, 1))           # 1320 rows
            want_d, want_i = est.kneighbors(x)
            sh = ShardedKNN(est)
            got_d, got_i = sh.kneighbors(x)
            np.testing.assert_array_equal(got_i, want_i)
            np.testing.assert_array_equal(got_d, want_d)
            xt = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device="cuda")
            cd, ci = sh.kneighbors_cyclic(xt, chunk_rows=500)   # 3 chunks, the last one ragged
            np.testing.assert_array_equal(ci.cpu().numpy(), want_i)
            np.testing.assert_array_equal(cd.cpu().numpy(), want_d)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["connectivity", "distance"])
def test_kneighbors_graph(E, moscow, mode):
    """The graph method scikit-learn's KNeighborsMixin gives the reference's RawKNNRegressor: CSR rows of the k
    neighbours (ones or distances), X=None excluding each row itself; an unknown mode raises scikit-learn's sentence."""
    from oracle import oracle as O

    X, y = moscow["X_train"], moscow["y_train"]
    est = E.RawKNNRegressor(n_neighbors=4).fit(X, y)
    for query, want in ((moscow["X_test"], O.kneighbors(X, moscow["X_test"], 4)), (None, O.kneighbors(X, None, 4))):
        g = est.kneighbors_graph(query, mode=mode)
        od, oi = want
        assert g.shape == (len(oi), len(X)) and g.nnz == oi.size
        dense = np.zeros(g.shape)
        np.put_along_axis(dense, oi, np.ones_like(od) if mode == "connectivity" else od, axis=1)
        np.testing.assert_array_equal(g.toarray(), dense)
    assert est.kneighbors_graph(moscow["X_test"], n_neighbors=2).nnz == 2 * len(moscow["X_test"])
    with pytest.raises(ValueError, match="Unsupported mode"):
        est.kneighbors_graph(moscow["X_test"], mode="nope")
