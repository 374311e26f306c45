#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the *reference itself*.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONPATH=/root/reference/src python tests/golden/make_golden.py

It imports lemma-osu/sknnr from /root/reference/src, runs its estimators on the
shipped Moscow Mountain / St. Joes and SWO ecoplot data and on the seeded
synthetic problems of ``sknnr_amd.synth``, and stores inputs/expected outputs as
small ``.npz`` files.  Only arrays are written (data, fitted matrices, neighbour
indices, distances, predictions, scores); no reference source travels.

``tests/golden/ref_regressions/*.npz`` are the reference's own regression data
files (raw, euclidean, mahalanobis, gnn, msn, and -- since round 2 -- randomForest and gbnn),
copied verbatim from /root/reference/tests/test_regressions/.
"""

from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import sknnr  # noqa: E402  (the reference)
from sklearn.model_selection import train_test_split  # noqa: E402
from sknnr import (  # noqa: E402
    EuclideanKNNRegressor,
    GBNNRegressor,
    GNNRegressor,
    RFNNRegressor,
    MahalanobisKNNRegressor,
    MSNRegressor,
    RawKNNRegressor,
)
from sknnr.datasets import load_moscow_stjoes, load_swo_ecoplot  # noqa: E402

from sknnr_amd import synth  # noqa: E402

assert sknnr.__file__.startswith("/root/reference/"), sknnr.__file__


def yaimpute_weights(d):
    return 1.0 / (1.0 + d)


def transformer_params(est) -> dict:
    out = {}
    tr = getattr(est, "transformer_", None)
    if tr is None:
        return out
    for name in ("mean_", "scale_", "env_center_", "projector_", "transform_"):
        if hasattr(tr, name):
            out["tr_" + name] = np.asarray(getattr(tr, name))
    sc = getattr(tr, "scaler_", None)
    if sc is not None:
        out["tr_scaler_mean_"] = np.asarray(sc.mean_)
        out["tr_scaler_scale_"] = np.asarray(sc.scale_)
    if hasattr(tr, "n_components_"):
        out["tr_n_components_"] = np.asarray(tr.n_components_)
    return out


def dump_dataset(name, ds):
    np.savez_compressed(
        os.path.join(ROOT, "sknnr_amd", "datasets", "data", name + ".npz"),
        index=np.asarray(ds.index, dtype=np.int64),
        data=np.asarray(ds.data, dtype=np.float64),
        target=np.asarray(ds.target, dtype=np.float64),
        feature_names=np.asarray(ds.feature_names),
        target_names=np.asarray(ds.target_names),
    )


ESTIMATORS = {
    "raw": (RawKNNRegressor, {}),
    "euclidean": (EuclideanKNNRegressor, {}),
    "mahalanobis": (MahalanobisKNNRegressor, {}),
    "gnn_full": (GNNRegressor, {}),
    "gnn_reduced": (GNNRegressor, {"n_components": 3}),
    "msn_full": (MSNRegressor, {}),
    "msn_reduced": (MSNRegressor, {"n_components": 3}),
}


def moscow_cases():
    X, y = load_moscow_stjoes(return_X_y=True, as_frame=True)
    X_train, X_test, y_train, y_test = train_test_split(
        X, y, train_size=0.8, shuffle=False
    )
    for name, (cls, kw) in ESTIMATORS.items():
        out = {}
        for k in (1, 5, 7):
            est = cls(n_neighbors=k, **kw).fit(X_train, y_train)
            reg = getattr(est, "regressor_", est)
            if k == 5:
                out.update(transformer_params(est))
                out["fit_method"] = np.asarray(reg._fit_method)
                out["n_features_in_"] = np.asarray(est.n_features_in_)
                if hasattr(est, "transformer_"):
                    out["Xt_train"] = est.transformer_.transform(X_train)
                    out["Xt_test"] = est.transformer_.transform(X_test)
                out["indep_pred_uniform"] = est.independent_prediction_
                out["indep_score_uniform"] = np.asarray(est.independent_score_)
                out["pred_tgt_uniform"] = est.predict(X_test)
                out["score_tgt_uniform"] = np.asarray(est.score(X_test, y_test))
            d, i = est.kneighbors()
            out[f"kn_ref_k{k}_dist"], out[f"kn_ref_k{k}_nn"] = d, i
            d, i = est.kneighbors(X_test)
            out[f"kn_tgt_k{k}_dist"], out[f"kn_tgt_k{k}_nn"] = d, i
            d, i = est.kneighbors(X_test, return_dataframe_index=True)
            out[f"kn_tgt_k{k}_ids"] = i
            if k == 5:
                d, i = est.kneighbors(X_test, use_deterministic_ordering=False)
                out["kn_tgt_k5_nd_dist"], out["kn_tgt_k5_nd_nn"] = d, i
                d, i = est.kneighbors(use_deterministic_ordering=False)
                out["kn_ref_k5_nd_dist"], out["kn_ref_k5_nd_nn"] = d, i
        for wname, w in (("distance", "distance"), ("yaimpute", yaimpute_weights)):
            for k in (5, 7):
                est = cls(n_neighbors=k, weights=w, **kw).fit(X_train, y_train)
                out[f"indep_pred_{wname}_k{k}"] = est.independent_prediction_
                out[f"indep_score_{wname}_k{k}"] = np.asarray(est.independent_score_)
                out[f"pred_tgt_{wname}_k{k}"] = est.predict(X_test)
        np.savez_compressed(os.path.join(HERE, f"moscow_{name}.npz"), **out)
        print("moscow", name, str(out["fit_method"]), len(out), "arrays")


def swo_case():
    """BASELINE config 1: MSNRegressor on load_swo_ecoplot(), k=5."""
    X, y = load_swo_ecoplot(return_X_y=True, as_frame=True)
    est = MSNRegressor(n_neighbors=5).fit(X, y)
    d, i = est.kneighbors()
    dq, iq = est.kneighbors(X)
    out = dict(
        indep_score=np.asarray(est.independent_score_),
        indep_pred_rows=est.independent_prediction_[::16],
        kn_ref_dist=d,
        kn_ref_nn=i.astype(np.int32),
        kn_self_dist=dq,
        kn_self_nn=iq.astype(np.int32),
        pred_rows=est.predict(X)[::16],
        fit_method=np.asarray(est.regressor_._fit_method),
        n_features_in_=np.asarray(est.n_features_in_),
        **transformer_params(est),
    )
    np.savez_compressed(os.path.join(HERE, "swo_msn_k5.npz"), **out)
    print("swo msn score", float(est.independent_score_), "D_t", est.n_features_in_)


def synthetic_raw_cases():
    """RawKNNRegressor (no transform) on seeded synthetic problems, incl. exact
    duplicate references (distance ties) and queries that are copies of
    references (zero distances)."""
    for d in (8, 16, 32, 64):
        for dup in (False, True):
            n_ref, n_q, t = 2048, 1024, 6
            x_ref, y, x_q = synth.make_problem(
                n_ref, n_q, d, t=t, n_dup_refs=96 if dup else 0,
                n_dup_queries=64 if dup else 0,
            )
            out = {}
            for k in (1, 5, 7):
                est = RawKNNRegressor(n_neighbors=k, weights="distance").fit(x_ref, y)
                dist, nn = est.kneighbors(x_q)
                out[f"tgt_k{k}_dist"], out[f"tgt_k{k}_nn"] = dist, nn.astype(np.int32)
                dist, nn = est.kneighbors()
                out[f"ref_k{k}_dist"], out[f"ref_k{k}_nn"] = dist, nn.astype(np.int32)
                out[f"pred_distance_k{k}"] = est.predict(x_q)
                out[f"indep_score_distance_k{k}"] = np.asarray(est.independent_score_)
                if k == 5:
                    out["fit_method"] = np.asarray(est._fit_method)
                    est_u = RawKNNRegressor(n_neighbors=k).fit(x_ref, y)
                    out["pred_uniform_k5"] = est_u.predict(x_q)
                    out["indep_pred_uniform_k5"] = est_u.independent_prediction_
                    out["indep_score_uniform_k5"] = np.asarray(est_u.independent_score_)
                    # brute forced (ArgKmin expanded formula) even when D <= 15
                    est_b = RawKNNRegressor(n_neighbors=k, algorithm="brute").fit(x_ref, y)
                    dist, nn = est_b.kneighbors(x_q)
                    out["tgt_k5_brute_dist"] = dist
                    out["tgt_k5_brute_nn"] = nn.astype(np.int32)
            tag = f"synth_raw_d{d}{'_dup' if dup else ''}"
            np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
            print(tag, str(out["fit_method"]))


def synthetic_estimator_cases():
    """Transformed estimators on the synthetic law (small versions of C2-C5)."""
    n_ref, n_q, d, t = 1500, 512, 16, 20
    cases = {
        "euclidean": (EuclideanKNNRegressor, dict(n_neighbors=5), "linear"),
        "mahalanobis": (MahalanobisKNNRegressor, dict(n_neighbors=5), "linear"),
        "gnn": (GNNRegressor, dict(n_neighbors=7, weights="distance"), "positive"),
        "msn": (MSNRegressor, dict(n_neighbors=1, n_components=8), "linear"),
    }
    for name, (cls, kw, kind) in cases.items():
        x_ref, y, x_q = synth.make_problem(n_ref, n_q, d, t=t, kind=kind)
        est = cls(**kw).fit(x_ref, y)
        dist, nn = est.kneighbors(x_q)
        out = dict(
            dist=dist,
            nn=nn.astype(np.int32),
            pred=est.predict(x_q),
            indep_score=np.asarray(est.independent_score_),
            indep_pred_rows=est.independent_prediction_[::8],
            fit_method=np.asarray(est.regressor_._fit_method),
            n_features_in_=np.asarray(est.n_features_in_),
            **transformer_params(est),
        )
        np.savez_compressed(os.path.join(HERE, f"synth_est_{name}.npz"), **out)
        print("synth_est", name, "D_t", est.n_features_in_, str(out["fit_method"]))


def synthetic_wide_estimator_cases():
    """The two BASELINE shapes the d = 16 fixtures above do not reach: GNN in 32 dimensions (C3) and
    Mahalanobis in 64 (C4), at fixture size."""
    cases = {
        "gnn_d32": (GNNRegressor, dict(n_neighbors=7, weights="distance"), "positive", 32),
        "mahalanobis_d64": (MahalanobisKNNRegressor, dict(n_neighbors=5), "linear", 64),
    }
    for name, (cls, kw, kind, d) in cases.items():
        x_ref, y, x_q = synth.make_problem(2000, 512, d, t=40, kind=kind)  # 40 targets: CCA keeps all 32 axes
        est = cls(**kw).fit(x_ref, y)
        dist, nn = est.kneighbors(x_q)
        out = dict(
            dist=dist, nn=nn.astype(np.int32), pred=est.predict(x_q),
            indep_score=np.asarray(est.independent_score_), indep_pred_rows=est.independent_prediction_[::8],
            fit_method=np.asarray(est.regressor_._fit_method), n_features_in_=np.asarray(est.n_features_in_),
            **transformer_params(est),
        )
        np.savez_compressed(os.path.join(HERE, f"synth_est_{name}.npz"), **out)
        print("synth_est", name, "D_t", est.n_features_in_, str(out["fit_method"]))


def tree_cases():
    """RFNN / GBNN on the Moscow data (the reference's regression configuration: random_state=42,
    k=5): node-id matrices, Hamming weights and every output of the hot path, so that the GPU search
    can be checked on exactly the reference's inputs without growing forests on the GPU box, plus a
    synthetic case with real-valued tree weights (GBNN, train_improvement) and more rows."""
    X, y = load_moscow_stjoes(return_X_y=True, as_frame=True)
    X_train, X_test, y_train, y_test = train_test_split(X, y, train_size=0.8, shuffle=False)
    for name, cls, kw in (("rfnn", RFNNRegressor, dict(random_state=42)),
                          ("gbnn", GBNNRegressor, dict(random_state=42)),
                          ("rfnn_weighted", RFNNRegressor, dict(random_state=42, n_estimators=20,
                                                                 forest_weights=np.arange(1, y.shape[1] + 1))),
                          ("gbnn_uniform", GBNNRegressor, dict(random_state=42, n_estimators=30,
                                                               tree_weighting_method="uniform"))):
        out = {}
        for k in (1, 5):
            est = cls(n_neighbors=k, **kw).fit(X_train, y_train)
            if k == 5:
                out["ids_train"] = est.transformer_.transform(X_train)
                out["ids_test"] = est.transformer_.transform(X_test)
                out["hamming_weights"] = est.hamming_weights_
                out["indep_pred_uniform"] = est.independent_prediction_
                out["indep_score_uniform"] = np.asarray(est.independent_score_)
                out["pred_tgt_uniform"] = est.predict(X_test)
                d, i = est.kneighbors(X_test, use_deterministic_ordering=False)
                out["kn_tgt_k5_nd_dist"], out["kn_tgt_k5_nd_nn"] = d, i
            d, i = est.kneighbors()
            out[f"kn_ref_k{k}_dist"], out[f"kn_ref_k{k}_nn"] = d, i
            d, i = est.kneighbors(X_test)
            out[f"kn_tgt_k{k}_dist"], out[f"kn_tgt_k{k}_nn"] = d, i
            d, i = est.kneighbors(X_test, return_dataframe_index=True)
            out[f"kn_tgt_k{k}_ids"] = i
        est = cls(n_neighbors=5, weights=yaimpute_weights, **kw).fit(X_train, y_train)
        out["indep_pred_yaimpute"] = est.independent_prediction_
        out["pred_tgt_yaimpute"] = est.predict(X_test)
        est = cls(n_neighbors=5, weights="distance", **kw).fit(X_train, y_train)
        out["pred_tgt_distance"] = est.predict(X_test)
        np.savez_compressed(os.path.join(HERE, f"moscow_{name}.npz"), **out)
        print("moscow", name, out["ids_train"].shape, "weights", len(out["hamming_weights"]))
    x_ref, yy, x_q = synth.make_problem(1200, 400, 8, t=3, kind="linear")
    est = GBNNRegressor(n_neighbors=5, n_estimators=25, random_state=0).fit(x_ref, yy)
    d, i = est.kneighbors(x_q)
    dr, ir = est.kneighbors()
    np.savez_compressed(os.path.join(HERE, "synth_gbnn.npz"), ids_ref=est.transformer_.transform(x_ref),
                        ids_q=est.transformer_.transform(x_q), hamming_weights=est.hamming_weights_, y=yy,
                        dist=d, nn=i.astype(np.int32), ref_dist=dr, ref_nn=ir.astype(np.int32), pred=est.predict(x_q))
    print("synth_gbnn", est.transformer_.transform(x_ref).shape)


def main():
    if "--only-new" in sys.argv:  # round 2 additions only (the earlier files stay byte for byte)
        synthetic_wide_estimator_cases()
        tree_cases()
        return
    dump_dataset("moscow_stjoes", load_moscow_stjoes())
    dump_dataset("swo_ecoplot", load_swo_ecoplot())
    moscow_cases()
    swo_case()
    synthetic_raw_cases()
    synthetic_estimator_cases()
    synthetic_wide_estimator_cases()
    tree_cases()


if __name__ == "__main__":
    main()
