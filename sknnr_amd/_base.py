"""Estimator surface of the MI355X backend: the reference's L3 layer re-stated over
:class:`sknnr_amd._engine.KNNEngine`.

Mirrors /root/reference/src/sknnr/_base.py -- same class names, constructor parameters,
method signatures, fitted attributes and error messages -- but no arithmetic of the
hot path happens in Python or scikit-learn: ``kneighbors`` / ``predict`` / the fit-time
independent prediction are launches of ``libsknnr_hip.so``.  There is no CPU fallback;
metrics other than Euclidean raise.

    RawKNNRegressor                  REF _base.py:43-182
    TransformedKNeighborsRegressor   REF _base.py:185-358
    YFitMixin                        REF _base.py:361-374
    OrdinationKNeighborsRegressor    REF _base.py:377-408
"""

from __future__ import annotations

import numbers
from abc import ABC, abstractmethod

import numpy as np
from sklearn.base import BaseEstimator, MultiOutputMixin, RegressorMixin
from sklearn.exceptions import NotFittedError
from sklearn.metrics import r2_score
from sklearn.utils.validation import _is_arraylike, check_is_fitted, validate_data

from . import _config, _native
from ._engine import KNNEngine, default_device, is_torch_cuda_tensor

# Element types query rows may keep on their way to the device (include/sknnr_hip.h, sknnr_dtype): the kernel that reads
# them widens to float64, exactly -- the reference's host-side conversion (validate_data(dtype=FLOAT_DTYPES) followed by
# float64 arithmetic, REF transformers/_cca_transformer.py:78-87) without the host pass and at the rows' own PCIe width.
# Anything else (int64, bool, float16, ...) is converted to float64 by validate_data as before.
_QUERY_DTYPES = [np.float64, np.float32, np.int16, np.uint16, np.uint8, np.int32]

_EUCLIDEAN_NAMES = {"euclidean", "l2"}
_ALGORITHMS = {"auto", "brute", "kd_tree", "ball_tree"}


def _effective_metric(metric, p, metric_params):
    """The Euclidean metric (SKL/neighbors/_base.py:429-472 maps minkowski/p=2 to 'euclidean') and the
    weighted Hamming metric of the tree-based estimators (REF _weighted_trees.py:53-59) exist on the
    device."""
    if metric == "hamming":
        extra = set(metric_params or {}) - {"w"}
        if extra:
            raise NotImplementedError(f"metric_params {sorted(extra)} are not supported with metric='hamming'")
        return "hamming"
    if metric_params:
        raise NotImplementedError(
            "metric_params are not supported by the MI355X backend (Euclidean metric only)")
    if callable(metric):
        raise NotImplementedError(
            "callable metrics are not supported by the MI355X backend (Euclidean metric only)")
    if metric == "minkowski":
        if p != 2:
            raise NotImplementedError(
                f"minkowski with p={p} is not supported by the MI355X backend (p must be 2)")
        return "euclidean"
    if metric in _EUCLIDEAN_NAMES:
        return "euclidean"
    raise NotImplementedError(
        f"metric={metric!r} is not supported by the MI355X backend (Euclidean metric only)")


def _resolve_fit_method(algorithm, n_ref, d, k):
    """Which of the reference's engines -- hence which float64 distance expression -- the
    ``algorithm`` setting selects (SKL/neighbors/_base.py:620-648)."""
    if algorithm not in _ALGORITHMS:
        raise ValueError(f"unrecognized algorithm: {algorithm!r}")
    if algorithm != "auto":
        return algorithm
    if d > 15 or (k is not None and k >= n_ref // 2):
        return "brute"
    return "kd_tree"


def replay_reference_selection(full, kk, call_rows, self_query, deterministic, decimals):
    """The reference's selection on full distance rows, line by line, with this host's numpy.

    ``full`` (n, n_fit): float64 distance rows (computed on the device).  ``kk``: neighbours searched (k, + 1 for the X=None
    path).  ``call_rows`` (n): each row's position in the whole call (for X=None also its own reference index).
    Steps: ``_kneighbors_reduce_func`` (SKL/neighbors/_base.py:733-760: argpartition, then argsort of the kept
    distances), the X=None self removal (SKL/neighbors/_base.py:936-963) and sknnr's deterministic reorder
    (REF src/sknnr/_base.py:166-175).  Returns ``(dist, idx)`` with k columns.
    """
    n = full.shape[0]
    sample_range = np.arange(n)[:, None]
    neigh = np.argpartition(full, kk - 1, axis=1)[:, :kk]
    neigh = neigh[sample_range, np.argsort(full[sample_range, neigh])]
    nd = full[sample_range, neigh]
    call_rows = np.asarray(call_rows, dtype=np.int64)
    if self_query:
        sample_mask = neigh != call_rows[:, None]
        dup_gr_nbrs = np.all(sample_mask, axis=1)
        sample_mask[:, 0][dup_gr_nbrs] = False
        neigh = np.reshape(neigh[sample_mask], (n, kk - 1))
        nd = np.reshape(nd[sample_mask], (n, kk - 1))
    if deterministic:
        row_scale = np.maximum(nd.max(axis=1, keepdims=True), 1.0)
        rounded = np.round(nd / row_scale, decimals=decimals)
        diff = np.abs(neigh - call_rows[:, None])
        order = np.lexsort((neigh, diff, rounded), axis=1)
        nd = np.take_along_axis(nd, order, axis=1)
        neigh = np.take_along_axis(neigh, order, axis=1)
    return nd, neigh


def _reraise(err, estimator):
    """Map a native failure to the exception the reference raises for the same input."""
    if err.code == _native.ERR_K_TOO_LARGE:
        raise ValueError(err.message) from None
    if err.code == _native.ERR_NONFINITE:
        # scikit-learn's own wording, estimator-specific second sentence included
        # (SKL/utils/validation.py _assert_all_finite)
        from sklearn.utils.validation import check_array

        bad = np.array([[np.nan if "NaN" in err.message else np.inf]])
        check_array(bad, estimator=estimator, input_name="X", ensure_min_features=1)
        raise ValueError(err.message) from None  # not reached
    raise err


class DFIndexCrosswalkMixin:
    """Capture of a dataframe's index at fit time (REF _base.py:23-30)."""

    def _set_dataframe_index_in(self, X) -> None:
        index = getattr(X, "index", None)
        if _is_arraylike(index):
            self.dataframe_index_in_ = np.asarray(index)


class RawKNNRegressor(DFIndexCrosswalkMixin, MultiOutputMixin, RegressorMixin, BaseEstimator):
    """k-nearest-neighbour regressor on the features as given, with sknnr's extras:
    dataframe-index crosswalk, fit-time independent prediction/score and deterministic
    neighbour ordering -- computed on the GPU.

    Same parameters as ``sklearn.neighbors.KNeighborsRegressor`` (REF _base.py:53-73);
    ``leaf_size`` and ``n_jobs`` are accepted and ignored (there is no tree and no thread
    pool), ``algorithm`` only selects which of the reference's two float64 distance
    expressions is reproduced.
    """

    DISTANCE_PRECISION_DECIMALS = 10

    def __init__(self, n_neighbors=5, *, weights="uniform", algorithm="auto", leaf_size=30, p=2,
                 metric="minkowski", metric_params=None, n_jobs=None):
        self.n_neighbors = n_neighbors
        self.weights = weights
        self.algorithm = algorithm
        self.leaf_size = leaf_size
        self.p = p
        self.metric = metric
        self.metric_params = metric_params
        self.n_jobs = n_jobs

    # -- fitting -------------------------------------------------------------------------
    def _check_params(self):
        if not isinstance(self.n_neighbors, numbers.Integral) or isinstance(self.n_neighbors, bool):
            raise TypeError(
                f"n_neighbors does not take {type(self.n_neighbors)} value, enter integer value")
        if self.n_neighbors <= 0:
            raise ValueError(f"Expected n_neighbors > 0. Got {self.n_neighbors}")
        if not (self.weights in (None, "uniform", "distance") or callable(self.weights)):
            raise ValueError(
                "weights not recognized: should be 'uniform', 'distance', or a callable function")
        self.effective_metric_ = _effective_metric(self.metric, self.p, self.metric_params)
        self.effective_metric_params_ = dict(self.metric_params or {}) if self.effective_metric_ == "hamming" else {}
        if self.effective_metric_ == "hamming" and self.algorithm not in ("auto", "brute"):
            raise ValueError(f"Metric 'hamming' not valid. Use sorted(sklearn.neighbors.VALID_METRICS['{self.algorithm}']) "
                             "to get valid options. Metric can also be a callable function.")

    def fit(self, X, y):
        """Store the reference rows and targets in HBM and compute the independent
        (leave-self-out) prediction and score (REF _base.py:104-109)."""
        self._set_dataframe_index_in(X)
        return self._fit_arrays(X, y, affine=None)

    def _fit_arrays(self, X, y, affine, device=None):
        self._check_params()
        X, y = validate_data(self, X, y, multi_output=True, order="C", dtype=np.float64,
                             ensure_all_finite=True, reset=True)
        self._y = y
        self._fit_X = X
        self.n_samples_fit_ = X.shape[0]
        self._fit_method = ("brute" if self.effective_metric_ == "hamming" else
                            _resolve_fit_method(self.algorithm, X.shape[0], X.shape[1], self.n_neighbors))
        if self.effective_metric_ == "hamming":
            w = self.effective_metric_params_.get("w")
            w = np.ones(X.shape[1]) if w is None else np.asarray(w, dtype=np.float64).reshape(-1)
            if w.size != X.shape[1]:
                raise ValueError(f"the Hamming weights have {w.size} entries for {X.shape[1]} columns")
            self._hamming_w = w
        self._affine = affine
        self._device = default_device() if device is None else device
        self._build_engine()
        self._set_independent_prediction_attributes(y)
        return self

    def _build_engine(self):
        y2 = self._y.reshape(-1, 1) if self._y.ndim == 1 else self._y
        self._engine = KNNEngine(self._fit_X, np.asarray(y2, dtype=np.float64), device=self._device)
        if getattr(self, "effective_metric_", "euclidean") == "hamming":
            self._engine.set_hamming_weights(self._hamming_w)
        if self._affine is not None:
            d_in, center, scale, proj = self._affine
            self._engine.set_affine(d_in, center, scale, proj)

    @property
    def engine_(self) -> KNNEngine:
        """The device engine; rebuilt lazily after unpickling."""
        check_is_fitted(self, "_fit_X")
        if getattr(self, "_engine", None) is None:
            self._build_engine()
        return self._engine

    def __getstate__(self):
        state = dict(super().__getstate__())
        state["_engine"] = None  # device handles do not pickle; see engine_
        return state

    def _formula(self) -> str:
        if getattr(self, "effective_metric_", "euclidean") == "hamming":
            return "hamming"
        return "direct" if self._fit_method == "kd_tree" else "expanded"

    def _set_independent_prediction_attributes(self, y) -> None:
        """REF _base.py:37-40: predict and score with X=None."""
        self.independent_prediction_ = self.predict(None)
        self.independent_score_ = float(r2_score(y, self.independent_prediction_))

    # -- queries -------------------------------------------------------------------------
    def _validate_query(self, X):
        if is_torch_cuda_tensor(X):
            if X.ndim != 2 or X.shape[1] != self.n_features_in_:
                raise ValueError(
                    f"X has {X.shape[1] if X.ndim == 2 else '?'} features, but {type(self).__name__} "
                    f"is expecting {self.n_features_in_} features as input.")
            return X
        # feature count / names / dtype here; finiteness is tested by the kernels that read the rows
        # (check_finite), not by a second pass over them on the host
        return validate_data(self, X, reset=False, order="C", dtype=_QUERY_DTYPES, ensure_all_finite=False)

    def _resolve_k(self, n_neighbors):
        if n_neighbors is None:
            return self.n_neighbors
        if not isinstance(n_neighbors, numbers.Integral) or isinstance(n_neighbors, bool):
            raise TypeError(
                "n_neighbors does not take %s value, enter integer value" % type(n_neighbors))
        if n_neighbors <= 0:
            raise ValueError("Expected n_neighbors > 0. Got %d" % n_neighbors)
        return int(n_neighbors)

    def _numpy_ties(self) -> bool:
        """RFNN / GBNN under ``hamming_tie_policy("numpy")``: exactly tied rows as the reference's argpartition keeps them."""
        return (getattr(self, "effective_metric_", "euclidean") == "hamming"
                and _config.get_hamming_tie_policy() == "numpy")

    def _kneighbors_hamming_numpy_ties(self, X, k, use_deterministic_ordering, row_offset, n_self_rows):
        """Weighted-Hamming neighbours with the REFERENCE's choice among exactly tied rows.

        The device answers every row (tied rows lowest index first).  A second device search for one neighbour more, on the
        raw rows, shows which queries have an exact tie that the choice depends on (across the last slot; without the
        deterministic reorder, anywhere among the kept rows).  For those queries the device returns the full float64
        distance row (``sknnr_hamming_distances``: the matrix the reference's brute search materialises) and the
        selection is replayed line by line with this host's numpy: ``_kneighbors_reduce_func``
        (SKL/neighbors/_base.py:733-760: argpartition, argsort), the X=None self removal (SKL/neighbors/_base.py:936-963)
        and sknnr's reorder (REF src/sknnr/_base.py:166-175) -- reached in the reference from
        REF src/sknnr/_weighted_trees.py:53-59, :139-140 and pinned by REF tests/test_regressions.py:125-195.
        """
        eng = self.engine_
        cuda_in = is_torch_cuda_tensor(X)
        X_host = X.cpu().numpy() if cuda_in else X
        self_query = X is None
        kk = k + (1 if self_query else 0)
        n_fit = self.n_samples_fit_
        dist, idx = eng.kneighbors(X_host, k, exclude_self=self_query, deterministic=use_deterministic_ordering,
                                   decimals=self.DISTANCE_PRECISION_DECIMALS, formula="hamming", row_offset=row_offset,
                                   n_self_rows=n_self_rows, check_finite=X is not None)
        nq = idx.shape[0]
        if nq:
            rows_q = self._fit_X[row_offset:row_offset + nq] if self_query else X_host
            probe = min(kk + 1, n_fit)
            pd, _ = eng.kneighbors(rows_q, probe, exclude_self=False, deterministic=False, formula="hamming")
            if use_deterministic_ordering:  # only the SET of kept rows can differ: a tie across the last slot
                flagged = pd[:, kk - 1] == pd[:, kk] if probe > kk else np.zeros(nq, dtype=bool)
            else:  # the order among equal distances is argpartition's too
                flagged = (pd[:, :-1] == pd[:, 1:]).any(axis=1) if probe > 1 else np.zeros(nq, dtype=bool)
            rows = np.flatnonzero(flagged)
            step = max(1, (256 << 20) // (8 * n_fit))  # distance rows of at most ~256 MB at a time
            for a in range(0, rows.size, step):
                sel = rows[a:a + step]
                full = eng.hamming_distances(None if self_query else X_host, sel + (row_offset if self_query else 0))
                dist[sel], idx[sel] = replay_reference_selection(full, kk, sel + row_offset, self_query,
                                                                  use_deterministic_ordering,
                                                                  self.DISTANCE_PRECISION_DECIMALS)
            self._last_numpy_tie_rows = int(rows.size)
        if cuda_in:
            import torch

            return torch.as_tensor(dist, device=X.device), torch.as_tensor(idx, device=X.device)
        return dist, idx

    def _kneighbors_engine(self, X, k, *, apply_affine, use_deterministic_ordering, row_offset=0,
                           n_self_rows=None, return_distance=True, out=None, owner=None):
        try:
            if self._numpy_ties() and out is None:
                return self._kneighbors_hamming_numpy_ties(X, k, use_deterministic_ordering, row_offset, n_self_rows)
            return self.engine_.kneighbors(
                X, k, exclude_self=X is None, deterministic=use_deterministic_ordering,
                decimals=self.DISTANCE_PRECISION_DECIMALS, formula=self._formula(),
                apply_affine=apply_affine, row_offset=row_offset, n_self_rows=n_self_rows,
                return_distance=return_distance, out=out, check_finite=X is not None)
        except _native.HipBackendError as err:
            _reraise(err, owner if owner is not None else self)

    def kneighbors(self, X=None, n_neighbors=None, return_distance=True, return_dataframe_index=False,
                   use_deterministic_ordering=True):
        """Neighbours of ``X`` (or of every fitted row, itself excluded, when ``X`` is None);
        same contract as REF _base.py:111-182.  CUDA tensors in give CUDA tensors out."""
        check_is_fitted(self, "_fit_X")
        k = self._resolve_k(n_neighbors)
        if X is not None:
            X = self._validate_query(X)
        dist, idx = self._kneighbors_engine(X, k, apply_affine=False,
                                            use_deterministic_ordering=use_deterministic_ordering)
        return self._finish_kneighbors(dist, idx, return_distance, return_dataframe_index)

    def kneighbors_graph(self, X=None, n_neighbors=None, mode="connectivity"):
        """Sparse (n_queries, n_samples_fit) graph of the k neighbours of every row: ones, or the distances
        with ``mode="distance"`` -- the method the reference's estimator inherits from scikit-learn
        (SKL/neighbors/_base.py, KNeighborsMixin.kneighbors_graph), over this class's ``kneighbors``."""
        from scipy.sparse import csr_matrix

        check_is_fitted(self, "_fit_X")
        k = self._resolve_k(n_neighbors)
        if mode == "connectivity":
            ind = self.kneighbors(X, k, return_distance=False)
            ind = ind.cpu().numpy() if is_torch_cuda_tensor(ind) else ind
            data = np.ones(ind.shape[0] * k)
        elif mode == "distance":
            data, ind = self.kneighbors(X, k, return_distance=True)
            if is_torch_cuda_tensor(ind):
                data, ind = data.cpu().numpy(), ind.cpu().numpy()
            data = np.ravel(data)
        else:
            raise ValueError(
                f'Unsupported mode, must be one of "connectivity", or "distance" but got "{mode}" instead')
        n_queries = ind.shape[0]
        indptr = np.arange(0, n_queries * k + 1, k)
        return csr_matrix((data, ind.ravel(), indptr), shape=(n_queries, self.n_samples_fit_))

    def _finish_kneighbors(self, dist, idx, return_distance, return_dataframe_index):
        if return_dataframe_index:
            msg = "Dataframe indexes can only be returned when fitted with a dataframe."
            check_is_fitted(self, "dataframe_index_in_", msg=msg)
            table = self.dataframe_index_in_
            if table.dtype.kind in "iu" and table.dtype.itemsize <= 8 and table.dtype != np.uint64:
                idx = self.engine_.crosswalk(idx, table.astype(np.int64, copy=False)).reshape(idx.shape)
                if not is_torch_cuda_tensor(idx):
                    idx = idx.astype(table.dtype, copy=False)
            else:  # labels that are not 64-bit integers (strings, ...) cannot live on the device
                host_idx = idx.cpu().numpy() if is_torch_cuda_tensor(idx) else idx
                idx = table[host_idx]
        return (dist, idx) if return_distance else idx

    def _predict_engine(self, X, *, apply_affine, row_offset=0, n_self_rows=None, owner=None):
        weights = None if self.weights is None else self.weights
        try:
            if self._numpy_ties():
                # the reference's predict() calls ITS kneighbors (deterministic ordering on): the same neighbours here,
                # then the reduction on the device (SKL/neighbors/_regression.py:224-268)
                dist, idx = self._kneighbors_hamming_numpy_ties(X, self.n_neighbors, True, row_offset, n_self_rows)
                cuda = is_torch_cuda_tensor(dist)
                if cuda:
                    dev = dist.device
                    dist, idx = dist.cpu().numpy(), idx.cpu().numpy()
                if callable(weights):
                    w = np.asarray(weights(dist), dtype=np.float64)
                    pred = self.engine_._index.predict_from_neighbors_host(dist, idx, w, _native.WEIGHTS_EXPLICIT)
                else:
                    mode = _native.WEIGHTS_DISTANCE if weights == "distance" else _native.WEIGHTS_UNIFORM
                    pred = self.engine_._index.predict_from_neighbors_host(dist, idx, None, mode)
                if cuda:
                    import torch

                    pred = torch.as_tensor(pred, device=dev)
                return pred.reshape(-1) if self._y.ndim == 1 else pred
            pred = self.engine_.predict(
                X, self.n_neighbors, "uniform" if weights is None else weights, exclude_self=X is None,
                deterministic=True, decimals=self.DISTANCE_PRECISION_DECIMALS, formula=self._formula(),
                apply_affine=apply_affine, row_offset=row_offset, n_self_rows=n_self_rows,
                check_finite=X is not None)
        except _native.HipBackendError as err:
            _reraise(err, owner if owner is not None else self)
        if self._y.ndim == 1:
            pred = pred.reshape(-1)
        return pred

    def predict(self, X):
        """Weighted mean of the neighbours' targets (SKL/neighbors/_regression.py:224-268);
        ``X=None`` predicts every fitted row from its neighbours, itself excluded."""
        check_is_fitted(self, "_fit_X")
        if X is not None:
            X = self._validate_query(X)
        return self._predict_engine(X, apply_affine=False)

    # -- streamed tiles (raster ingestion; REF docs/pages/usage.md:101-128) --------------------
    def _stream_tiles(self, tiles, validate, k, *, apply_affine, weights, return_distance,
                      use_deterministic_ordering, out, owner):
        """Push host tiles through one native query stream.  Returns (dist, idx, pred) arrays over all
        pushed rows (pieces of ``out`` when given, else concatenated)."""
        eng = self.engine_
        want_pred = weights is not None
        t_cols = eng.t
        if self._numpy_ties():
            # the reference's choice among tied rows is made on the host, per call: tile by tile, positions carried
            parts, row = [], 0
            for tile in tiles:
                tile = validate(tile)
                if tile.shape[0] == 0:
                    continue
                if want_pred:
                    parts.append((None, None, np.reshape(self._predict_engine(tile, apply_affine=apply_affine, row_offset=row,
                                                                               owner=owner), (tile.shape[0], -1))))
                else:
                    d_, i_ = self._kneighbors_engine(tile, k, apply_affine=apply_affine, row_offset=row,
                                                     use_deterministic_ordering=use_deterministic_ordering, owner=owner)
                    parts.append((d_, i_, None))
                row += tile.shape[0]
            cat = lambda j, cols, dt: (np.concatenate([p_[j] for p_ in parts]) if parts  # noqa: E731
                                       else np.empty((0, cols), dtype=dt))
            res = (cat(0, k, np.float64) if return_distance and not want_pred else None,
                   None if want_pred else cat(1, k, np.int64), cat(2, t_cols, np.float64) if want_pred else None)
            if out is not None:
                for dst, src in zip(out, res):
                    if dst is not None and src is not None:
                        dst[:row] = src
                res = tuple(None if (dst is None or src is None) else dst[:row] for dst, src in zip(out, res))
            return res
        o_dist, o_idx, o_pred = out if out is not None else (None, None, None)
        pieces = []
        row = 0
        stream = None
        stream_dtype = None
        try:
            for tile in tiles:
                if is_torch_cuda_tensor(tile):
                    raise TypeError("streamed tiles are host arrays (they travel through the pinned "
                                    "PCIe pipeline); pass CUDA tensors to kneighbors() / predict()")
                tile = validate(tile)
                n = tile.shape[0]
                if n == 0:
                    continue
                if stream is None:
                    # the element type of the first tile is the stream's (narrow rasters travel at their own width)
                    code = eng.query_dtype_code(tile, self._formula())
                    stream_dtype = tile.dtype if code else np.dtype(np.float64)
                    stream = eng.open_stream(k, weights=weights, want_dist=return_distance,
                                             deterministic=use_deterministic_ordering,
                                             decimals=self.DISTANCE_PRECISION_DECIMALS, formula=self._formula(),
                                             apply_affine=apply_affine, check_finite=True, query_dtype=code)
                if tile.dtype != stream_dtype:
                    if stream_dtype != np.float64:
                        raise ValueError(f"the tiles of one streamed call must share an element type: got {tile.dtype} "
                                         f"after {stream_dtype}")
                    tile = np.ascontiguousarray(tile, dtype=np.float64)

                def window(arr, cols, dtype):
                    if arr is None:
                        return None
                    w = arr[row:row + n]
                    if w.shape != (n, cols) or w.dtype != dtype or not w.flags.c_contiguous:
                        raise ValueError(f"out arrays must be C-contiguous, {np.dtype(dtype)}, with "
                                         f"{cols} columns and at least {row + n} rows")
                    return w

                got = stream.push(tile, out_idx=window(o_idx, k, np.int64),
                                  out_dist=window(o_dist, k, np.float64) if return_distance else None,
                                  out_pred=window(o_pred, t_cols, np.float64) if want_pred else None,
                                  need_idx=not want_pred)
                if out is None:
                    pieces.append(got)
                row += n
            if stream is not None:
                done, stream = stream, None
                done.close()
        except _native.HipBackendError as err:
            _reraise(err, owner if owner is not None else self)
        finally:
            if stream is not None:  # an error on the way: free the native stream without masking it
                try:
                    stream.close()
                except _native.HipBackendError:
                    pass
        if out is not None:
            trim = lambda a: None if a is None else a[:row]  # noqa: E731
            return trim(o_dist) if return_distance else None, trim(o_idx), trim(o_pred) if want_pred else None
        cat = lambda i, cols, dt: (np.concatenate([p[i] for p in pieces]) if pieces  # noqa: E731
                                   else np.empty((0, cols), dtype=dt))
        return (cat(1, k, np.float64) if return_distance else None,
                None if want_pred else cat(0, k, np.int64),
                cat(2, t_cols, np.float64) if want_pred else None)

    def kneighbors_chunks(self, tiles, n_neighbors=None, return_distance=True, return_dataframe_index=False,
                          use_deterministic_ordering=True, out=None):
        """``kneighbors`` over an iterable of host tiles ``(n_i, n_features)`` -- windows of a raster,
        slices of a ``numpy.memmap`` -- as ONE logical call: the copy-in / kernels / copy-out pipeline
        stays full across tiles and row positions count over all tiles, so the result equals
        ``kneighbors(np.concatenate(tiles))`` bit for bit.  ``out=(dist, idx)`` (``dist`` may be None):
        preallocated arrays (e.g. memmaps) that receive the rows in order."""
        check_is_fitted(self, "_fit_X")
        k = self._resolve_k(n_neighbors)
        o = None if out is None else (out[0], out[1], None)
        dist, idx, _ = self._stream_tiles(tiles, self._validate_query, k, apply_affine=False, weights=None,
                                          return_distance=return_distance,
                                          use_deterministic_ordering=use_deterministic_ordering, out=o, owner=None)
        return self._finish_chunks(dist, idx, return_distance, return_dataframe_index)

    def _finish_chunks(self, dist, idx, return_distance, return_dataframe_index):
        if return_dataframe_index:
            msg = "Dataframe indexes can only be returned when fitted with a dataframe."
            check_is_fitted(self, "dataframe_index_in_", msg=msg)
            table = self.dataframe_index_in_
            if table.dtype == np.int64:
                step = 1 << 22  # in place, block by block: idx may be a memmap larger than memory
                for a in range(0, idx.shape[0], step):
                    idx[a:a + step] = table[idx[a:a + step]]
            else:
                idx = table[idx]
        return (dist, idx) if return_distance else idx

    def predict_chunks(self, tiles, out=None):
        """``predict`` over an iterable of host tiles as one streamed call; ``out``: preallocated
        ``(n_rows, n_targets)`` float64 array (e.g. a memmap)."""
        check_is_fitted(self, "_fit_X")
        return self._predict_chunks(tiles, self._validate_query, apply_affine=False, out=out, owner=None)

    def _predict_chunks(self, tiles, validate, *, apply_affine, out, owner):
        weights = "uniform" if self.weights is None else self.weights
        if callable(weights):  # a Python callable runs between the search and the reduction: tile by tile
            # (the reorder's second key is the row's position in the WHOLE call: carry it from tile to tile)
            preds, row = [], 0
            for t in tiles:
                t = validate(t)
                preds.append(self._predict_engine(t, apply_affine=apply_affine, row_offset=row, owner=owner))
                row += t.shape[0]
            pred = np.concatenate([p.reshape(len(p), -1) for p in preds]) if preds else np.empty((0, self.engine_.t))
            if out is not None:
                out[:len(pred)] = pred
                pred = out[:len(pred)]
        else:
            o = None if out is None else (None, None, out.reshape(out.shape[0], -1))
            _, _, pred = self._stream_tiles(tiles, validate, self.n_neighbors, apply_affine=apply_affine,
                                            weights=weights, return_distance=False,
                                            use_deterministic_ordering=True, out=o, owner=owner)
        return pred.reshape(-1) if self._y.ndim == 1 else pred

    def score(self, X, y, sample_weight=None):
        """R^2 of ``predict(X)`` (``X`` may be None as in REF _base.py:40)."""
        pred = self.predict(X)
        if is_torch_cuda_tensor(pred):
            pred = pred.cpu().numpy()
        return float(r2_score(y, pred, sample_weight=sample_weight))

    def __sklearn_tags__(self):
        tags = super().__sklearn_tags__()
        tags.input_tags.sparse = False
        return tags


class TransformedKNeighborsRegressor(BaseEstimator, ABC):
    """kNN regressors that search in a transformed feature space (REF _base.py:185-358).

    ``fit`` learns the transformer on the host, maps the training rows through the GPU
    affine kernel, and installs the same map in the engine so that ``kneighbors`` /
    ``predict`` take *untransformed* rows and transform them inside the launch.
    """

    def __init__(self, n_neighbors=5, *, weights="uniform", algorithm="auto", leaf_size=30, p=2,
                 metric="minkowski", metric_params=None, n_jobs=None):
        self.n_neighbors = n_neighbors
        self.weights = weights
        self.algorithm = algorithm
        self.leaf_size = leaf_size
        self.p = p
        self.metric = metric
        self.metric_params = metric_params
        self.n_jobs = n_jobs

    @abstractmethod
    def _get_transformer(self):
        """The (unfitted) transformer that defines the feature space."""

    def _set_fitted_transformer(self, X, y) -> None:
        self.transformer_ = self._get_transformer().fit(X, y)

    def _get_additional_regressor_init_kwargs(self) -> dict:
        return {}

    def _transform_X(self, X):
        """Host-side transform, kept for API parity (REF _base.py:236-239); the hot path does
        not use it -- queries are transformed on the device."""
        check_is_fitted(self, "transformer_")
        return self.transformer_.transform(X) if X is not None else X

    def fit(self, X, y):
        validate_data(self, X=X, y=y, ensure_all_finite=True, multi_output=True)
        self._set_fitted_transformer(X, y)

        device = default_device()
        # Affine feature spaces are applied on the device (fit rows here, query rows inside the launch);
        # tree-node spaces (scikit-learn forests) are evaluated on the host and the device searches node ids.
        self._device_affine = hasattr(self.transformer_, "affine_params")
        if self._device_affine:
            center, scale, proj = self.transformer_.affine_params()
            X_arr = np.ascontiguousarray(
                validate_data(self.transformer_, X=X, reset=False, dtype=np.float64), dtype=np.float64)
            X_transformed = _native.affine_transform_host(X_arr, center, scale, proj, device=device)
            affine = (X_arr.shape[1], center, scale, proj)
        else:
            X_transformed = self.transformer_.transform(X)
            affine = None

        kwargs = {
            "n_neighbors": self.n_neighbors, "weights": self.weights, "algorithm": self.algorithm,
            "leaf_size": self.leaf_size, "p": self.p, "metric": self.metric,
            "metric_params": self.metric_params, "n_jobs": self.n_jobs,
        }
        kwargs.update(self._get_additional_regressor_init_kwargs())
        self.regressor_ = RawKNNRegressor(**kwargs)
        self.regressor_._fit_arrays(X_transformed, y, affine=affine, device=device)
        self.regressor_._set_dataframe_index_in(X)

        self.n_features_in_ = self.regressor_.n_features_in_
        self.independent_prediction_ = self.regressor_.independent_prediction_
        self.independent_score_ = self.regressor_.independent_score_
        if hasattr(self.regressor_, "dataframe_index_in_"):
            self.dataframe_index_in_ = self.regressor_.dataframe_index_in_
        return self

    def _validate_raw_query(self, X):
        """Same checks the transformer's ``transform`` applies (feature names/count, finiteness)
        without transforming on the host -- or, for host-side feature spaces, the transform itself."""
        check_is_fitted(self, "transformer_")
        if not getattr(self, "_device_affine", True):
            if is_torch_cuda_tensor(X):
                X = X.cpu().numpy()
            return np.ascontiguousarray(self.transformer_.transform(X), dtype=np.float64)
        if is_torch_cuda_tensor(X):
            d_in = self.regressor_.engine_.d_in
            if X.ndim != 2 or X.shape[1] != d_in:
                raise ValueError(f"X has {X.shape[-1]} features, but {type(self).__name__} is expecting "
                                 f"{d_in} features as input.")
            return X
        # finiteness is tested on the device by the kernel that reads the rows (check_finite)
        return validate_data(self.transformer_, X=X, reset=False, dtype=_QUERY_DTYPES, order="C",
                             ensure_all_finite=False)

    def kneighbors(self, X=None, n_neighbors=None, return_distance=True, return_dataframe_index=False,
                   use_deterministic_ordering=True):
        """REF _base.py:285-344."""
        check_is_fitted(self, "regressor_")
        reg = self.regressor_
        k = reg._resolve_k(n_neighbors)
        if X is not None:
            X = self._validate_raw_query(X)
        dist, idx = reg._kneighbors_engine(X, k, apply_affine=X is not None and self._device_affine,
                                           use_deterministic_ordering=use_deterministic_ordering,
                                           owner=self.transformer_)
        return reg._finish_kneighbors(dist, idx, return_distance, return_dataframe_index)

    def predict(self, X):
        """REF _base.py:346-348 (``X=None``: the independent prediction, as the reference's
        ``_transform_X(None)`` passes None through to the regressor)."""
        check_is_fitted(self, "regressor_")
        if X is not None:
            X = self._validate_raw_query(X)
        return self.regressor_._predict_engine(X, apply_affine=X is not None and self._device_affine,
                                               owner=self.transformer_)

    def kneighbors_chunks(self, tiles, n_neighbors=None, return_distance=True, return_dataframe_index=False,
                          use_deterministic_ordering=True, out=None):
        """``kneighbors`` over an iterable of untransformed host tiles as one streamed call (see
        :meth:`RawKNNRegressor.kneighbors_chunks`); each tile is transformed on the device."""
        check_is_fitted(self, "regressor_")
        reg = self.regressor_
        k = reg._resolve_k(n_neighbors)
        o = None if out is None else (out[0], out[1], None)
        dist, idx, _ = reg._stream_tiles(tiles, self._validate_raw_query, k, apply_affine=self._device_affine, weights=None,
                                         return_distance=return_distance,
                                         use_deterministic_ordering=use_deterministic_ordering, out=o,
                                         owner=self.transformer_)
        return reg._finish_chunks(dist, idx, return_distance, return_dataframe_index)

    def predict_chunks(self, tiles, out=None):
        """``predict`` over an iterable of untransformed host tiles as one streamed call."""
        check_is_fitted(self, "regressor_")
        return self.regressor_._predict_chunks(tiles, self._validate_raw_query, apply_affine=self._device_affine,
                                               out=out, owner=self.transformer_)

    def score(self, X, y):
        """REF _base.py:350-352."""
        pred = self.predict(X)
        if is_torch_cuda_tensor(pred):
            pred = pred.cpu().numpy()
        return float(r2_score(y, pred))

    def __sklearn_tags__(self):
        tags = super().__sklearn_tags__()  # (as REF _base.py:354-358: a plain BaseEstimator that rejects sparse input)
        tags.input_tags.sparse = False
        return tags


class YFitMixin(TransformedKNeighborsRegressor):
    """Optional ``y_fit`` that only the transformer sees (REF _base.py:361-374)."""

    def _set_fitted_transformer(self, X, y) -> None:
        y_fit = self.y_fit_ if self.y_fit_ is not None else y
        self.transformer_ = self._get_transformer().fit(X, y_fit)

    def fit(self, X, y, y_fit=None):
        self.y_fit_ = y_fit
        return super().fit(X, y)


class OrdinationKNeighborsRegressor(TransformedKNeighborsRegressor, ABC):
    """Transformed regressors with an ``n_components`` knob (REF _base.py:377-408)."""

    def __init__(self, n_neighbors=5, *, n_components=None, weights="uniform", algorithm="auto",
                 leaf_size=30, p=2, metric="minkowski", metric_params=None, n_jobs=None):
        super().__init__(n_neighbors=n_neighbors, weights=weights, algorithm=algorithm,
                         leaf_size=leaf_size, p=p, metric=metric, metric_params=metric_params,
                         n_jobs=n_jobs)
        self.n_components = n_components


__all__ = [
    "RawKNNRegressor",
    "TransformedKNeighborsRegressor",
    "YFitMixin",
    "OrdinationKNeighborsRegressor",
    "NotFittedError",
]
