"""Device engine behind the estimators: one :class:`KNNEngine` per fitted regressor.

It owns the native index handle (reference rows, targets and the query-time affine map
resident in HBM) and moves queries/results as numpy arrays (staged through PCIe by the
library) or as PyTorch-ROCm CUDA tensors (zero copy, launched on torch's current stream).
All arithmetic happens in ``libsknnr_hip.so``; nothing here computes distances.
"""

from __future__ import annotations

import os

import numpy as np

from . import _native

__all__ = ["KNNEngine", "default_device", "set_default_device", "is_torch_cuda_tensor"]

_default_device: int | None = None


def set_default_device(index: int | None) -> None:
    """Pin new engines to HIP device ``index`` (None = follow LOCAL_RANK / torch)."""
    global _default_device
    _default_device = index


def default_device() -> int:
    """Device for new engines: explicit override, else LOCAL_RANK (one process per GPU
    under ``torch.distributed.run``), else torch's current device, else 0."""
    if _default_device is not None:
        return _default_device
    import torch

    if torch.cuda.is_available():
        n = torch.cuda.device_count()
        if "LOCAL_RANK" in os.environ and n > 0:
            return int(os.environ["LOCAL_RANK"]) % n
        return torch.cuda.current_device()
    return 0


def is_torch_cuda_tensor(x) -> bool:
    mod = type(x).__module__
    return mod.startswith("torch") and hasattr(x, "is_cuda") and bool(x.is_cuda)


_WEIGHT_MODES = {"uniform": _native.WEIGHTS_UNIFORM, None: _native.WEIGHTS_UNIFORM,
                 "distance": _native.WEIGHTS_DISTANCE}


class KNNEngine:
    """HBM-resident kNN index + the kneighbors / predict launches.

    Parameters
    ----------
    fit_X : (n_ref, d) float64, the *transformed* reference rows (the reference's ``_fit_X``)
    y : (n_ref, t) targets or None
    """

    def __init__(self, fit_X, y=None, device: int | None = None):
        self.device = default_device() if device is None else int(device)
        self._index = _native.Index(fit_X, y, device=self.device)
        self.n_ref = self._index.n_ref
        self.d = self._index.d
        self.t = self._index.t
        self.d_in = self.d
        self.has_affine = False

    def close(self):
        self._index.close()

    def set_affine(self, d_in, center=None, scale=None, proj=None):
        self._index.set_affine(int(d_in), center, scale, proj)
        self.d_in = int(d_in)
        self.has_affine = True

    def set_hamming_weights(self, w):
        """Per-column weights of the weighted-Hamming search over tree node ids (``formula="hamming"``)."""
        self._index.set_hamming_weights(w)

    def stats(self) -> dict:
        return self._index.stats()

    def reset_stats(self):
        self._index.reset_stats()

    # ------------------------------------------------------------------------------------
    def _opts(self, k, *, exclude_self, deterministic, decimals, formula, apply_affine,
              weight_mode=_native.WEIGHTS_UNIFORM, row_offset=0, check_finite=False, query_dtype=0):
        return self._index.make_opts(
            k, exclude_self=exclude_self, deterministic=deterministic, decimals=decimals,
            formula={"direct": _native.FORMULA_DIRECT, "hamming": _native.FORMULA_HAMMING}.get(
                formula, _native.FORMULA_EXPANDED),
            apply_affine=apply_affine, weight_mode=weight_mode, row_offset=row_offset,
            check_finite=check_finite, query_dtype=query_dtype)

    def query_dtype_code(self, X, formula="expanded") -> int:
        """The sknnr_dtype under which the rows of ``X`` can be handed to the library as they are (float32, int16, uint16,
        uint8, int32: widened by the kernel that reads them, exactly), or 0 = float64 (convert first): narrow rows need
        the MFMA envelope (d <= 128) and a Euclidean formula."""
        if X is None or formula == "hamming" or self.d > 128:
            return 0
        return _native.dtype_code(X.dtype) or 0

    def _as_device_rows(self, X, apply_affine, query_dtype=0):
        """A contiguous CUDA tensor on the engine's device with the expected columns: float64, or the narrower element
        type ``query_dtype`` names (then left as it is)."""
        import torch

        if (query_dtype == 0 and X.dtype != torch.float64) or not X.is_contiguous():
            X = (X if query_dtype else X.to(torch.float64)).contiguous()
        if X.device.index != self.device:
            raise ValueError(f"X is on cuda:{X.device.index}, the engine on cuda:{self.device}")
        self._check_columns(X, apply_affine)
        return X

    def _check_columns(self, X, apply_affine):
        want = self.d_in if apply_affine else self.d
        if X.shape[1] != want:
            raise ValueError(f"X has {X.shape[1]} features, the engine expects {want}")

    def kneighbors(self, X, k, *, exclude_self=False, deterministic=True, decimals=10,
                   formula="expanded", apply_affine=False, row_offset=0, n_self_rows=None,
                   return_distance=True, out=None, check_finite=False):
        """Neighbours of the rows of ``X`` (numpy -> numpy, torch.cuda -> torch.cuda), or of
        the reference rows ``[row_offset, row_offset + n_self_rows)`` when ``X`` is None.
        ``out=(dist, idx)``: contiguous float64 / int64 CUDA tensors of shape ``(nq, k)`` to write
        into (torch.cuda input only), e.g. this rank's slot of an all-gather buffer.
        ``check_finite``: the kernels that read ``X`` also test it for NaN / infinity and the call
        raises ``HipBackendError(ERR_NONFINITE)`` (for CUDA tensors this synchronises the stream)."""
        qdt = self.query_dtype_code(X, formula)
        opts = self._opts(k, exclude_self=exclude_self, deterministic=deterministic,
                          decimals=decimals, formula=formula,
                          apply_affine=apply_affine and X is not None, row_offset=row_offset,
                          check_finite=check_finite and X is not None, query_dtype=qdt)
        if X is None:
            if not exclude_self:
                raise ValueError("X=None requires exclude_self=True")
            nq = self.n_ref - row_offset if n_self_rows is None else int(n_self_rows)
            return self._index.kneighbors_host(None, opts, nq=nq, return_distance=return_distance)
        if is_torch_cuda_tensor(X):
            import torch

            X = self._as_device_rows(X, apply_affine, qdt)
            nq = X.shape[0]
            if out is not None:
                dist, idx = out
                for t_, dt_ in ((dist, torch.float64), (idx, torch.int64)):
                    if t_ is None:
                        continue
                    if (tuple(t_.shape) != (nq, k) or t_.dtype != dt_ or not t_.is_contiguous()
                            or t_.device != X.device):
                        raise ValueError(f"out tensors must be contiguous ({nq}, {k}) {dt_} on {X.device}")
                if idx is None:
                    raise ValueError("out needs an index tensor")
            else:
                idx = torch.empty((nq, k), dtype=torch.int64, device=X.device)
                dist = torch.empty((nq, k), dtype=torch.float64, device=X.device) if return_distance else None
            if nq:
                stream = torch.cuda.current_stream(X.device).cuda_stream
                self._index.kneighbors_device(X.data_ptr(), nq, opts,
                                              dist.data_ptr() if dist is not None else 0,
                                              idx.data_ptr(), stream)
                if check_finite:
                    self._index.check_finite(stream)
            return dist, idx
        if out is not None:
            raise ValueError("out= is only supported for torch.cuda inputs")
        X = np.ascontiguousarray(X) if qdt else np.ascontiguousarray(X, dtype=np.float64)
        self._check_columns(X, apply_affine)
        return self._index.kneighbors_host(X, opts, return_distance=return_distance)

    def predict(self, X, k, weights="uniform", *, exclude_self=False, deterministic=True, decimals=10,
                formula="expanded", apply_affine=False, row_offset=0, n_self_rows=None, check_finite=False):
        """Weighted multi-output mean of the neighbours' targets."""
        if self.t < 1:
            raise ValueError("the engine was built without targets")
        if callable(weights):
            # A Python callable cannot run on the device: find the neighbours on the GPU, let
            # the callable map the (nq, k) distances to weights on the host, reduce on the GPU.
            dist, idx = self.kneighbors(X, k, exclude_self=exclude_self, deterministic=deterministic,
                                        decimals=decimals, formula=formula, apply_affine=apply_affine,
                                        row_offset=row_offset, n_self_rows=n_self_rows, check_finite=check_finite)
            if is_torch_cuda_tensor(dist):
                import torch

                w = weights(dist)
                w = torch.as_tensor(w, dtype=torch.float64, device=dist.device).contiguous()
                pred = torch.empty((dist.shape[0], self.t), dtype=torch.float64, device=dist.device)
                stream = torch.cuda.current_stream(dist.device).cuda_stream
                self._index.predict_from_neighbors_device(dist.data_ptr(), idx.data_ptr(), w.data_ptr(),
                                                          dist.shape[0], k, _native.WEIGHTS_EXPLICIT,
                                                          pred.data_ptr(), stream)
                return pred
            w = np.asarray(weights(dist), dtype=np.float64)
            if w.shape != dist.shape:
                raise ValueError("the weights callable must return an array shaped like its input")
            return self._index.predict_from_neighbors_host(dist, idx, w, _native.WEIGHTS_EXPLICIT)
        if weights not in _WEIGHT_MODES:
            raise ValueError(f"weights not recognized: should be 'uniform', 'distance', or a callable; got {weights!r}")
        mode = _WEIGHT_MODES[weights]
        qdt = self.query_dtype_code(X, formula)
        opts = self._opts(k, exclude_self=exclude_self, deterministic=deterministic, decimals=decimals,
                          formula=formula, apply_affine=apply_affine and X is not None,
                          weight_mode=mode, row_offset=row_offset, check_finite=check_finite and X is not None,
                          query_dtype=qdt)
        if X is None:
            nq = self.n_ref - row_offset if n_self_rows is None else int(n_self_rows)
            return self._index.predict_host(None, opts, nq=nq)
        if is_torch_cuda_tensor(X):
            import torch

            X = self._as_device_rows(X, apply_affine, qdt)
            nq = X.shape[0]
            pred = torch.empty((nq, self.t), dtype=torch.float64, device=X.device)
            if nq:
                stream = torch.cuda.current_stream(X.device).cuda_stream
                self._index.predict_device(X.data_ptr(), nq, opts, pred.data_ptr(), 0, 0, stream)
                if check_finite:
                    self._index.check_finite(stream)
            return pred
        X = np.ascontiguousarray(X) if qdt else np.ascontiguousarray(X, dtype=np.float64)
        self._check_columns(X, apply_affine)
        return self._index.predict_host(X, opts)

    # ---- reference-sharded search (sknnr_amd.distributed.RefShardedKNN) ------------------------------------
    def shard_candidates(self, X, kk, *, formula="expanded", apply_affine=False, index_offset=0, check_finite=False):
        """This engine's rows are one SHARD of a reference set: the ``kk`` nearest of them for every row of ``X`` as raw
        candidates -- values of the formula (squared distances) ascending by (value, index), indices +
        ``index_offset`` -- numpy in / numpy out or torch.cuda in / torch.cuda out."""
        opts = self._opts(kk, exclude_self=False, deterministic=False, decimals=10, formula=formula,
                          apply_affine=apply_affine, check_finite=check_finite)
        if is_torch_cuda_tensor(X):
            import torch

            X = self._as_device_rows(X, apply_affine)
            nq = X.shape[0]
            val = torch.empty((nq, kk), dtype=torch.float64, device=X.device)
            idx = torch.empty((nq, kk), dtype=torch.int64, device=X.device)
            if nq:
                stream = torch.cuda.current_stream(X.device).cuda_stream
                self._index.shard_candidates_device(X.data_ptr(), nq, opts, index_offset, val.data_ptr(), idx.data_ptr(), stream)
                if check_finite:
                    self._index.check_finite(stream)
            return val, idx
        X = np.ascontiguousarray(X, dtype=np.float64)
        self._check_columns(X, apply_affine)
        return self._index.shard_candidates_host(X, opts, index_offset)

    def merge_shards(self, X, k, shard_val, shard_idx, *, exclude_self=False, deterministic=True, decimals=10,
                     formula="expanded", apply_affine=False, row_offset=0, n_self_rows=None):
        """Merge the gathered candidates ``(n_shards, nq, k + exclude_self)`` of all shards into the call's final
        ``(dist, idx)``; this engine holds ALL reference rows and re-scans the rows whose merged answer is not unique."""
        opts = self._opts(k, exclude_self=exclude_self, deterministic=deterministic, decimals=decimals, formula=formula,
                          apply_affine=apply_affine and X is not None, row_offset=row_offset)
        if is_torch_cuda_tensor(shard_val):
            import torch

            n_shards, nq = shard_val.shape[0], shard_val.shape[1]
            Xd = None if X is None else self._as_device_rows(X, apply_affine)
            dist = torch.empty((nq, k), dtype=torch.float64, device=shard_val.device)
            idx = torch.empty((nq, k), dtype=torch.int64, device=shard_val.device)
            sv, si = shard_val.contiguous(), shard_idx.contiguous()
            if nq:
                stream = torch.cuda.current_stream(shard_val.device).cuda_stream
                self._index.merge_shards_device(0 if Xd is None else Xd.data_ptr(), nq, opts, n_shards, sv.data_ptr(),
                                                si.data_ptr(), dist.data_ptr(), idx.data_ptr(), stream)
            return dist, idx
        nq = shard_val.shape[1] if X is None else None
        if X is not None:
            X = np.ascontiguousarray(X, dtype=np.float64)
            self._check_columns(X, apply_affine)
        return self._index.merge_shards_host(X, opts, shard_val, shard_idx, nq=nq)

    def open_stream(self, k, *, weights=None, want_dist=True, deterministic=True, decimals=10,
                    formula="expanded", apply_affine=False, row_offset=0, check_finite=False, query_dtype=0):
        """A :class:`sknnr_amd._native.QueryStream` over host tiles: ``push(tile)`` keeps the PCIe
        pipeline full across tiles and carries the global row offset.  ``weights`` (``"uniform"`` /
        ``"distance"``) also asks for predictions."""
        want_pred = weights is not None
        if want_pred and weights not in _WEIGHT_MODES:
            raise ValueError("a stream predicts with 'uniform' or 'distance' weights only")
        opts = self._opts(k, exclude_self=False, deterministic=deterministic, decimals=decimals,
                          formula=formula, apply_affine=apply_affine,
                          weight_mode=_WEIGHT_MODES[weights] if want_pred else _native.WEIGHTS_UNIFORM,
                          row_offset=row_offset, check_finite=check_finite, query_dtype=query_dtype)
        return self._index.open_stream(opts, want_dist=want_dist, want_pred=want_pred)

    def hamming_distances(self, X, rows=None):
        """Full weighted-Hamming distance rows of ``X[rows]`` (``X`` None: of the fitted rows) from the device."""
        return self._index.hamming_distances_host(X, rows)

    def crosswalk(self, idx, table):
        """``table[idx]`` for int64 dataframe ids (REF _base.py:177-180)."""
        if is_torch_cuda_tensor(idx):
            import torch

            tab = torch.as_tensor(np.ascontiguousarray(table, dtype=np.int64), device=idx.device)
            out = torch.empty_like(idx)
            stream = torch.cuda.current_stream(idx.device).cuda_stream
            _native.crosswalk_device(tab.data_ptr(), tab.numel(), idx.data_ptr(), idx.numel(),
                                     out.data_ptr(), self.device, stream)
            torch.cuda.current_stream(idx.device).synchronize()  # tab dies with this frame
            return out
        return _native.crosswalk_host(table, idx, self.device)
