"""CPU oracle for the kneighbors()/predict() hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package ``sknnr_amd`` never does (a test
enforces that).  Parity status: **pinned** -- see the header of ``knn_oracle.c``.

The heavy arithmetic (distance scan, heap, sort) is the C restatement in
``knn_oracle.c``; the cheap post-steps are numpy and cite the reference lines
they follow (REF = /root/reference, SKL = site-packages/sklearn, 1.7.2).
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libknn_oracle.so")
_lib = None

_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> str:
    """Compile knn_oracle.c with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "knn_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libknn_oracle.so"], check=True,
                       capture_output=True)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    with open("/proc/cpuinfo") as fh:
        if " fma" not in fh.read():
            raise RuntimeError("oracle is built with -mfma; this host has no FMA")
    build()
    lib = ctypes.CDLL(_LIB_PATH)
    lib.oracle_num_threads.restype = ctypes.c_int
    lib.oracle_affine.argtypes = [
        _f64p, ctypes.c_int64, ctypes.c_int32, _f64p, _f64p, _f64p, ctypes.c_int32, _f64p,
    ]
    lib.oracle_row_norms.argtypes = [_f64p, ctypes.c_int64, ctypes.c_int32, _f64p]
    for name in ("oracle_argkmin_expanded", "oracle_argkmin_direct"):
        getattr(lib, name).argtypes = [
            _f64p, ctypes.c_int64, _f64p, ctypes.c_int64, ctypes.c_int32,
            ctypes.c_int32, ctypes.c_int32, _f64p, _i64p,
        ]
    lib.oracle_argkmin_hamming.argtypes = [
        _f64p, ctypes.c_int64, _f64p, ctypes.c_int64, ctypes.c_int32, _f64p, ctypes.c_int32, _f64p, _i64p,
    ]
    _lib = lib
    return lib


def num_threads() -> int:
    return int(_load().oracle_num_threads())


def _c64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ=_f64p):
    return None if a is None else a.ctypes.data_as(typ)


def affine(X, center=None, scale=None, P=None) -> np.ndarray:
    """``((X - center) / scale) @ P`` in float64; any of the three may be None.

    Follows ``StandardScaler.transform`` (SKL/preprocessing/_data.py:1057-1098),
    ``CCATransformer.transform`` (REF/src/sknnr/transformers/_cca_transformer.py:87),
    ``CCorATransformer.transform`` (_ccora_transformer.py:70) and
    ``MahalanobisTransformer.transform`` (_mahalanobis_transformer.py:55).
    """
    X = _c64(X)
    n, d_in = X.shape
    center = None if center is None else _c64(center)
    scale = None if scale is None else _c64(scale)
    P = None if P is None else _c64(P)
    d_out = d_in if P is None else P.shape[1]
    out = np.empty((n, d_out), dtype=np.float64)
    _load().oracle_affine(_ptr(X), n, d_in, _ptr(center), _ptr(scale), _ptr(P), d_out,
                          _ptr(out))
    return out


def row_norms(X) -> np.ndarray:
    X = _c64(X)
    out = np.empty(X.shape[0], dtype=np.float64)
    _load().oracle_row_norms(_ptr(X), X.shape[0], X.shape[1], _ptr(out))
    return out


def argkmin(Q, R, k: int, formula: str = "expanded", squared: bool = False):
    """k smallest distances of each row of Q to the rows of R.

    ``formula='expanded'`` is the ArgKmin/brute arithmetic, ``'direct'`` the
    kd_tree arithmetic (see knn_oracle.c).  Returns ``(dist float64, idx int64)``.
    """
    Q, R = _c64(Q), _c64(R)
    nq, d = Q.shape
    nr = R.shape[0]
    if R.shape[1] != d:
        raise ValueError("feature mismatch")
    if not 1 <= k <= nr:
        raise ValueError(f"k={k} out of range for n_ref={nr}")
    dist = np.empty((nq, k), dtype=np.float64)
    idx = np.empty((nq, k), dtype=np.int64)
    fn = {"expanded": "oracle_argkmin_expanded", "direct": "oracle_argkmin_direct"}[formula]
    getattr(_load(), fn)(_ptr(Q), nq, _ptr(R), nr, d, k, int(squared), _ptr(dist),
                         _ptr(idx, _i64p))
    return dist, idx


def argkmin_hamming(Q, R, w, k: int):
    """k smallest weighted-Hamming distances (see ``oracle_argkmin_hamming`` in knn_oracle.c); rows
    ordered by (distance, index).  Q, R: node-id matrices (any integer / float dtype)."""
    Q, R, w = _c64(Q), _c64(R), _c64(w).reshape(-1)
    nq, d = Q.shape
    nr = R.shape[0]
    if R.shape[1] != d or w.size != d:
        raise ValueError("feature mismatch")
    if not 1 <= k <= nr:
        raise ValueError(f"k={k} out of range for n_ref={nr}")
    dist = np.empty((nq, k), dtype=np.float64)
    idx = np.empty((nq, k), dtype=np.int64)
    _load().oracle_argkmin_hamming(_ptr(Q), nq, _ptr(R), nr, d, _ptr(w), k, _ptr(dist), _ptr(idx, _i64p))
    return dist, idx


def kneighbors_hamming(fit_ids, ids=None, w=None, k: int = 5, deterministic: bool = True, decimals: int = 10,
                       row_offset: int = 0):
    """``RawKNNRegressor.kneighbors`` of the RFNN / GBNN estimators: weighted Hamming distance on node
    ids, X=None drop-self, sknnr's reorder (REF _weighted_trees.py:53-59, _base.py:111-182)."""
    fit_ids = _c64(fit_ids)
    n_fit = fit_ids.shape[0]
    if ids is None:
        dist, idx = argkmin_hamming(fit_ids, fit_ids, w, k + 1)
        dist, idx = drop_self(dist, idx)
    else:
        dist, idx = argkmin_hamming(ids, fit_ids, w, k)
    if deterministic:
        dist, idx = deterministic_reorder(dist, idx, decimals, row_offset)
    return dist, idx


def fit_method(n_ref: int, d: int, k: int, algorithm: str = "auto") -> str:
    """Which engine the reference's ``algorithm`` setting resolves to
    (SKL/neighbors/_base.py:620-648, Euclidean metric)."""
    if algorithm != "auto":
        return algorithm
    if d > 15 or k >= n_ref // 2:
        return "brute"
    return "kd_tree"


def drop_self(dist: np.ndarray, idx: np.ndarray, row_offset: int = 0):
    """X=None post-step: remove each row's own index from its k+1 neighbours, or the
    first column when the row is not among them (SKL/neighbors/_base.py:936-963)."""
    n, k1 = idx.shape
    rows = np.arange(row_offset, row_offset + n)[:, None]
    keep = idx != rows
    all_other = keep.all(axis=1)
    keep[all_other, 0] = False
    return dist[keep].reshape(n, k1 - 1), idx[keep].reshape(n, k1 - 1)


def deterministic_reorder(dist, idx, decimals: int = 10, row_offset: int = 0):
    """sknnr's tie-break (REF/src/sknnr/_base.py:166-175): sort each row by
    (distance / max(rowmax, 1) rounded to `decimals`, |idx - row|, idx).
    ``row_offset`` is the global position of row 0 of this block in the call."""
    n = len(idx)
    scale = np.maximum(dist.max(axis=1, keepdims=True), 1.0)
    key0 = np.round(dist / scale, decimals=decimals)
    key1 = np.abs(idx - np.arange(row_offset, row_offset + n)[:, None])
    order = np.lexsort((idx, key1, key0), axis=1)
    return np.take_along_axis(dist, order, axis=1), np.take_along_axis(idx, order, axis=1)


def kneighbors(fit_X, X=None, k: int = 5, formula: str = "expanded",
               deterministic: bool = True, decimals: int = 10, row_offset: int = 0):
    """``RawKNNRegressor.kneighbors`` on already-transformed features
    (REF/src/sknnr/_base.py:111-182 over SKL/neighbors/_base.py:763-963)."""
    fit_X = _c64(fit_X)
    n_fit = fit_X.shape[0]
    if X is None:
        if k + 1 > n_fit:
            raise ValueError(
                f"Expected n_neighbors < n_samples_fit, but n_neighbors = {k}, "
                f"n_samples_fit = {n_fit}, n_samples = {n_fit}"
            )
        dist, idx = argkmin(fit_X, fit_X, k + 1, formula)
        dist, idx = drop_self(dist, idx)
    else:
        X = _c64(X)
        if k > n_fit:
            raise ValueError(
                f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, "
                f"n_samples_fit = {n_fit}, n_samples = {X.shape[0]}"
            )
        dist, idx = argkmin(X, fit_X, k, formula)
    if deterministic:
        dist, idx = deterministic_reorder(dist, idx, decimals, row_offset)
    return dist, idx


def shard_candidates(fit_X_shard, X, kk: int, formula: str = "expanded", index_offset: int = 0):
    """One reference shard's answer to every query row, as the merge wants it: the kk smallest values of the
    formula (squared distances) ascending by (value, index), indices + ``index_offset``.  The analogue of one
    thread's heaps in scikit-learn's parallel-on-Y strategy
    (SKL/metrics/_pairwise_distances_reduction/_argkmin.pyx.tp:200-236)."""
    val, idx = argkmin(X, fit_X_shard, kk, formula, squared=True)
    order = np.lexsort((idx, val), axis=1)
    return np.take_along_axis(val, order, 1), np.take_along_axis(idx, order, 1) + int(index_offset)


def merge_shards(fit_X, X, shard_val, shard_idx, k: int, formula: str = "expanded", deterministic: bool = True,
                 decimals: int = 10, row_offset: int = 0):
    """Merge of the shards' candidates into the call's answer (``X`` None: the X=None path, ``shard_val`` then
    holds k + 1 candidates per shard).  The merged list is the answer when it is unique -- smallest (value, index)
    first, which is what one heap over all rows returns when no exact tie decides anything -- and a row with a tie
    across the last slot (seen in the merged list, or possibly hidden behind a shard whose list ends at the
    boundary value), or, without deterministic ordering, with equal values among its kept rows, or whose own index
    is not among its k + 1, is answered by the full replay of the reference's heap (``argkmin`` over all rows):
    SKL/.../_argkmin.pyx.tp:237-261 (_parallel_on_Y_synchronize) has the same structure."""
    fit_X = _c64(fit_X)
    self_rows = X is None
    kk = k + (1 if self_rows else 0)
    Xq = fit_X[row_offset:row_offset + shard_val.shape[1]] if self_rows else _c64(X)
    G, nq, _ = shard_val.shape
    val = np.empty((nq, kk))
    idx = np.empty((nq, kk), dtype=np.int64)
    replay = []
    for q in range(nq):
        v = shard_val[:, q, :].ravel()
        i = shard_idx[:, q, :].ravel()
        order = np.lexsort((i, v))
        v, i = v[order], i[order]
        unique = True
        if formula == "expanded":
            vk = v[kk - 1]
            unique = not (len(v) > kk and v[kk] == vk)
            unique = unique and not any(shard_val[g, q, kk - 1] == vk for g in range(G))  # a full list ending at vk may hide rows
            if unique and not deterministic:
                unique = not (np.diff(v[:kk]) == 0).any()
            if unique and self_rows:
                unique = bool((i[:kk] == row_offset + q).any())
        if unique:
            val[q], idx[q] = v[:kk], i[:kk]
        else:
            replay.append(q)
    if replay:
        rv, ri = argkmin(Xq[replay], fit_X, kk, formula, squared=True)
        val[replay], idx[replay] = rv, ri
    if self_rows:
        val, idx = drop_self(val, idx, row_offset)
    dist = np.sqrt(val)
    if deterministic:
        dist, idx = deterministic_reorder(dist, idx, decimals, row_offset)
    return dist, idx, len(replay)


def get_weights(dist, weights):
    """SKL/neighbors/_base.py:81-124."""
    if weights in (None, "uniform"):
        return None
    if isinstance(weights, str) and weights == "distance":
        with np.errstate(divide="ignore"):
            w = 1.0 / dist
        inf_mask = np.isinf(w)
        inf_row = inf_mask.any(axis=1)
        w[inf_row] = inf_mask[inf_row]
        return w
    return weights(dist)


def predict(y, dist, idx, weights="uniform"):
    """SKL/neighbors/_regression.py:224-268."""
    y = np.asarray(y)
    y2 = y.reshape(-1, 1) if y.ndim == 1 else y
    w = get_weights(dist, weights)
    if w is None:
        pred = np.mean(y2[idx], axis=1)
    else:
        pred = np.empty((idx.shape[0], y2.shape[1]), dtype=np.float64)
        den = np.sum(w, axis=1)
        for j in range(y2.shape[1]):
            pred[:, j] = np.sum(y2[idx, j] * w, axis=1) / den
    return pred.ravel() if y.ndim == 1 else pred


def r2_uniform(y_true, y_pred) -> float:
    """R^2, uniform average over outputs (SKL/metrics/_regression.py r2_score with
    force_finite=True)."""
    y_true = np.asarray(y_true, dtype=np.float64)
    y_pred = np.asarray(y_pred, dtype=np.float64)
    if y_true.ndim == 1:
        y_true, y_pred = y_true[:, None], y_pred[:, None]
    num = ((y_true - y_pred) ** 2).sum(axis=0)
    den = ((y_true - y_true.mean(axis=0)) ** 2).sum(axis=0)
    score = np.ones(y_true.shape[1])
    ok = den != 0
    score[ok] = 1.0 - num[ok] / den[ok]
    score[(num != 0) & ~ok] = 0.0
    return float(score.mean())
