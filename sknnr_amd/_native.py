"""ctypes binding of the C ABI in ``include/sknnr_hip.h``.

The HIP library is the only engine: there is no CPU fallback.  Loading fails loudly
when ``libsknnr_hip.so`` is missing (build it with ``python -m sknnr_amd._build``) and
every compute call fails loudly when no MI355X is visible.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int32, c_int64, c_void_p

import numpy as np

from . import _build

MEM_HOST = 0
MEM_DEVICE = 1
FORMULA_EXPANDED = 0
FORMULA_DIRECT = 1
FORMULA_HAMMING = 2
WEIGHTS_UNIFORM = 0
WEIGHTS_DISTANCE = 1
WEIGHTS_EXPLICIT = 2

ERR_INVALID = -1
ERR_K_TOO_LARGE = -2
ERR_NO_TARGETS = -3
ERR_UNSUPPORTED = -4
ERR_HIP = -5
ERR_NO_DEVICE = -6
ERR_NONFINITE = -7
ABI_VERSION = 5

# sknnr_dtype (include/sknnr_hip.h): query rows narrower than float64 are widened by the kernel that reads them
DTYPE_CODES = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.int16): 2, np.dtype(np.uint16): 3,
               np.dtype(np.uint8): 4, np.dtype(np.int32): 5}


def dtype_code(dt) -> int | None:
    """The sknnr_dtype of a numpy / torch dtype, or None when the rows must be converted to float64 first."""
    try:
        return DTYPE_CODES.get(np.dtype(dt))
    except TypeError:  # a torch dtype
        name = str(dt).replace("torch.", "")
        try:
            return DTYPE_CODES.get(np.dtype(name))
        except TypeError:
            return None


# every symbol include/sknnr_hip.h declares (checked by tests/test_cabi.py)
EXPORTED_SYMBOLS = (
    "sknnr_device_count",
    "sknnr_abi_version",
    "sknnr_last_error",
    "sknnr_index_create",
    "sknnr_index_destroy",
    "sknnr_index_set_affine",
    "sknnr_index_set_hamming_weights",
    "sknnr_affine_transform",
    "sknnr_index_shape",
    "sknnr_get_stats",
    "sknnr_reset_stats",
    "sknnr_check_finite",
    "sknnr_kneighbors",
    "sknnr_predict",
    "sknnr_predict_from_neighbors",
    "sknnr_hamming_distances",
    "sknnr_shard_candidates",
    "sknnr_merge_shards",
    "sknnr_stream_begin",
    "sknnr_stream_push",
    "sknnr_stream_flush",
    "sknnr_stream_end",
    "sknnr_crosswalk",
    "sknnr_debug_coarse_matrix",
)


class QueryOpts(ctypes.Structure):
    _fields_ = [
        ("n_neighbors", c_int32),
        ("exclude_self", c_int32),
        ("deterministic", c_int32),
        ("decimals", c_int32),
        ("formula", c_int32),
        ("apply_affine", c_int32),
        ("weight_mode", c_int32),
        ("check_finite", c_int32),
        ("query_dtype", c_int32),
        ("reserved_", c_int32),
        ("row_offset", c_int64),
    ]


class Stats(ctypes.Structure):
    _fields_ = [
        ("queries", c_int64),
        ("coarse_queries", c_int64),
        ("exact_fallbacks", c_int64),
        ("exact_only_queries", c_int64),
        ("last_kernel_ms", c_double),
        ("last_coarse_ms", c_double),
        ("total_kernel_ms", c_double),
        ("total_coarse_ms", c_double),
        ("timed_calls", c_int64),
        ("coarse_rows_timed", c_int64),
        ("mfma_executed_ratio", c_double),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class HipBackendError(RuntimeError):
    """A call into libsknnr_hip.so failed; ``code`` is the sknnr_status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[sknnr_hip {code}] {message}")
        self.code = code
        self.message = message


_lib = None


def library_path() -> str:
    """The in-tree build, or the file named by SKNNR_HIP_LIBRARY (development variants)."""
    return os.environ.get("SKNNR_HIP_LIBRARY") or _build.LIB_PATH


def load(build_if_missing: bool = False):
    """Load libsknnr_hip.so (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm is the carrier of device memory and streams, and it bundles its own HIP
    # runtime (SONAME libamdhip64.so.7).  Importing it FIRST makes the dynamic loader bind
    # libsknnr_hip.so to that same runtime; loading ours first would put a second HIP/HSA
    # runtime in the process and torch would then see no device.
    import torch  # noqa: F401

    path = library_path()
    if not os.path.exists(path):
        if build_if_missing:
            _build.build()
        else:
            raise ImportError(
                f"{path} is missing: the MI355X backend has not been built. "
                "Run `python -m sknnr_amd._build` (needs hipcc); there is no CPU fallback."
            )
    lib = ctypes.CDLL(path)
    if lib.sknnr_abi_version() != ABI_VERSION:
        raise ImportError(
            f"{path} has ABI version {lib.sknnr_abi_version()}, this package needs {ABI_VERSION}: "
            "rebuild it with `python -m sknnr_amd._build --force`")
    vp = c_void_p
    lib.sknnr_device_count.restype = c_int32
    lib.sknnr_abi_version.restype = c_int32
    lib.sknnr_last_error.restype = c_char_p
    lib.sknnr_index_create.argtypes = [vp, c_int64, c_int32, vp, c_int32, c_int32, POINTER(vp)]
    lib.sknnr_index_destroy.argtypes = [vp]
    lib.sknnr_index_destroy.restype = None
    lib.sknnr_index_set_affine.argtypes = [vp, c_int32, vp, vp, vp]
    lib.sknnr_index_set_hamming_weights.argtypes = [vp, vp, c_int32]
    lib.sknnr_affine_transform.argtypes = [vp, c_int64, c_int32, vp, vp, vp, c_int32, vp, c_int32]
    lib.sknnr_index_shape.argtypes = [vp, POINTER(c_int64), POINTER(c_int32), POINTER(c_int32),
                                      POINTER(c_int32), POINTER(c_int32)]
    lib.sknnr_get_stats.argtypes = [vp, POINTER(Stats)]
    lib.sknnr_reset_stats.argtypes = [vp]
    lib.sknnr_check_finite.argtypes = [vp, vp]
    lib.sknnr_stream_begin.argtypes = [vp, POINTER(QueryOpts), c_int32, c_int32, POINTER(vp)]
    lib.sknnr_stream_push.argtypes = [vp, vp, c_int64, vp, vp, vp]
    lib.sknnr_stream_flush.argtypes = [vp]
    lib.sknnr_stream_end.argtypes = [vp, POINTER(c_int64)]
    lib.sknnr_kneighbors.argtypes = [vp, vp, c_int64, POINTER(QueryOpts), vp, vp, c_int32, vp]
    lib.sknnr_predict.argtypes = [vp, vp, c_int64, POINTER(QueryOpts), vp, vp, vp, c_int32, vp]
    lib.sknnr_predict_from_neighbors.argtypes = [vp, vp, vp, vp, c_int64, c_int32, c_int32, vp,
                                                 c_int32, vp]
    lib.sknnr_hamming_distances.argtypes = [vp, vp, c_int64, vp, c_int64, vp, c_int32, vp]
    lib.sknnr_shard_candidates.argtypes = [vp, vp, c_int64, POINTER(QueryOpts), c_int64, vp, vp, c_int32, vp]
    lib.sknnr_merge_shards.argtypes = [vp, vp, c_int64, POINTER(QueryOpts), c_int32, vp, vp, vp, vp, c_int32, vp]
    lib.sknnr_crosswalk.argtypes = [vp, c_int64, vp, c_int64, vp, c_int32, c_int32, vp]
    lib.sknnr_debug_coarse_matrix.argtypes = [vp, vp, c_int64, vp, vp, POINTER(c_double),
                                              POINTER(c_double)]
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        raise HipBackendError(code, load().sknnr_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    return int(load().sknnr_device_count())


def _host_ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _c_f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _c_rows(a, opts):
    """Query rows as the call will read them: C-contiguous, of the element type ``opts.query_dtype`` names (the caller set
    it from the array's dtype: Index.make_opts(query_dtype=...)), else float64."""
    if a is None:
        return None
    for dt, code in DTYPE_CODES.items():
        if code == opts.query_dtype:
            return np.ascontiguousarray(a, dtype=dt)
    raise ValueError(f"unknown query_dtype {opts.query_dtype}")


class Index:
    """Owner of one ``sknnr_index*``.  Inputs/outputs are numpy arrays (host) or raw
    device pointers (ints) -- see :mod:`sknnr_amd._engine` for the torch-tensor layer."""

    def __init__(self, ref, y=None, device: int = 0):
        lib = load()
        ref = _c_f64(ref)
        if ref.ndim != 2:
            raise ValueError("ref must be 2-D")
        y2 = None
        if y is not None:
            y2 = _c_f64(y)
            if y2.ndim == 1:
                y2 = y2.reshape(-1, 1)
            if y2.shape[0] != ref.shape[0]:
                raise ValueError("y and ref row counts differ")
        self.n_ref, self.d = ref.shape
        self.t = 0 if y2 is None else y2.shape[1]
        self.device = device
        self.d_in = self.d
        self._h = c_void_p()
        check(lib.sknnr_index_create(_host_ptr(ref), self.n_ref, self.d, _host_ptr(y2), self.t,
                                     device, byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load().sknnr_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("index is closed")
        return self._h

    def set_affine(self, d_in: int, center=None, scale=None, proj=None):
        center, scale, proj = _c_f64(center), _c_f64(scale), _c_f64(proj)
        if proj is not None and proj.shape != (d_in, self.d):
            raise ValueError(f"proj must be ({d_in}, {self.d}), got {proj.shape}")
        check(load().sknnr_index_set_affine(self.handle, d_in, _host_ptr(center), _host_ptr(scale),
                                            _host_ptr(proj)))
        self.d_in = d_in

    def set_hamming_weights(self, w):
        w = _c_f64(w).reshape(-1)
        check(load().sknnr_index_set_hamming_weights(self.handle, _host_ptr(w), w.size))

    def stats(self) -> dict:
        st = Stats()
        check(load().sknnr_get_stats(self.handle, byref(st)))
        return st.as_dict()

    def reset_stats(self):
        check(load().sknnr_reset_stats(self.handle))

    @staticmethod
    def make_opts(k, exclude_self=False, deterministic=True, decimals=10, formula=FORMULA_EXPANDED,
                  apply_affine=False, weight_mode=WEIGHTS_UNIFORM, row_offset=0,
                  check_finite=False, query_dtype=0) -> QueryOpts:
        return QueryOpts(int(k), int(bool(exclude_self)), int(bool(deterministic)), int(decimals),
                         int(formula), int(bool(apply_affine)), int(weight_mode), int(bool(check_finite)),
                         int(query_dtype), 0, int(row_offset))

    def check_finite(self, stream=0) -> None:
        """Poll the non-finite-input flag of device-memory calls made with ``check_finite``
        (synchronises ``stream``); raises :class:`HipBackendError` (``ERR_NONFINITE``)."""
        check(load().sknnr_check_finite(self.handle, c_void_p(stream or None)))

    def open_stream(self, opts: QueryOpts, want_dist=True, want_pred=False) -> "QueryStream":
        return QueryStream(self, opts, want_dist, want_pred)

    # ---- host (numpy) entry points --------------------------------------------------------
    def kneighbors_host(self, q, opts: QueryOpts, nq=None, return_distance=True):
        q = _c_rows(q, opts)
        if q is not None:
            nq = q.shape[0]
        k = opts.n_neighbors
        idx = np.empty((nq, k), dtype=np.int64)
        dist = np.empty((nq, k), dtype=np.float64) if return_distance else None
        check(load().sknnr_kneighbors(self.handle, _host_ptr(q), nq, byref(opts), _host_ptr(dist),
                                      _host_ptr(idx), MEM_HOST, None))
        return dist, idx

    def predict_host(self, q, opts: QueryOpts, nq=None, return_neighbors=False):
        q = _c_rows(q, opts)
        if q is not None:
            nq = q.shape[0]
        k = opts.n_neighbors
        pred = np.empty((nq, self.t), dtype=np.float64)
        dist = idx = None
        if return_neighbors:
            dist = np.empty((nq, k), dtype=np.float64)
            idx = np.empty((nq, k), dtype=np.int64)
        check(load().sknnr_predict(self.handle, _host_ptr(q), nq, byref(opts), _host_ptr(pred),
                                   _host_ptr(dist), _host_ptr(idx), MEM_HOST, None))
        return (pred, dist, idx) if return_neighbors else pred

    def predict_from_neighbors_host(self, dist, idx, w, weight_mode):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        dist, w = _c_f64(dist), _c_f64(w)
        nq, k = idx.shape
        pred = np.empty((nq, self.t), dtype=np.float64)
        check(load().sknnr_predict_from_neighbors(self.handle, _host_ptr(dist), _host_ptr(idx),
                                                  _host_ptr(w), nq, k, int(weight_mode),
                                                  _host_ptr(pred), MEM_HOST, None))
        return pred

    # ---- device-pointer entry points (ints from tensor.data_ptr()) ------------------------
    def kneighbors_device(self, q_ptr, nq, opts: QueryOpts, dist_ptr, idx_ptr, stream=0):
        check(load().sknnr_kneighbors(self.handle, c_void_p(q_ptr or None), nq, byref(opts),
                                      c_void_p(dist_ptr or None), c_void_p(idx_ptr), MEM_DEVICE,
                                      c_void_p(stream or None)))

    def predict_device(self, q_ptr, nq, opts: QueryOpts, pred_ptr, dist_ptr=0, idx_ptr=0, stream=0):
        check(load().sknnr_predict(self.handle, c_void_p(q_ptr or None), nq, byref(opts),
                                   c_void_p(pred_ptr), c_void_p(dist_ptr or None),
                                   c_void_p(idx_ptr or None), MEM_DEVICE, c_void_p(stream or None)))

    def predict_from_neighbors_device(self, dist_ptr, idx_ptr, w_ptr, nq, k, weight_mode, pred_ptr,
                                      stream=0):
        check(load().sknnr_predict_from_neighbors(
            self.handle, c_void_p(dist_ptr or None), c_void_p(idx_ptr), c_void_p(w_ptr or None), nq, k,
            int(weight_mode), c_void_p(pred_ptr), MEM_DEVICE, c_void_p(stream or None)))

    def hamming_distances_host(self, q, rows=None):
        """Full weighted-Hamming distance rows ``(len(rows), n_ref)`` of the query rows ``q[rows]`` (``q`` None: of the
        index's own rows), computed on the device in the reference's float64 arithmetic (sknnr_hamming_distances)."""
        q = _c_f64(q)
        nq = self.n_ref if q is None else q.shape[0]
        rows = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
        n_rows = nq if rows is None else rows.size
        out = np.empty((n_rows, self.n_ref), dtype=np.float64)
        check(load().sknnr_hamming_distances(self.handle, _host_ptr(q), nq, _host_ptr(rows), n_rows, _host_ptr(out),
                                             MEM_HOST, None))
        return out

    # ---- reference-sharded search (include/sknnr_hip.h) -------------------------------------------
    def shard_candidates_host(self, q, opts: QueryOpts, index_offset=0):
        q = _c_f64(q)
        nq, k = q.shape[0], opts.n_neighbors
        val = np.empty((nq, k), dtype=np.float64)
        idx = np.empty((nq, k), dtype=np.int64)
        check(load().sknnr_shard_candidates(self.handle, _host_ptr(q), nq, byref(opts), int(index_offset),
                                            _host_ptr(val), _host_ptr(idx), MEM_HOST, None))
        return val, idx

    def shard_candidates_device(self, q_ptr, nq, opts: QueryOpts, index_offset, val_ptr, idx_ptr, stream=0):
        check(load().sknnr_shard_candidates(self.handle, c_void_p(q_ptr), nq, byref(opts), int(index_offset),
                                            c_void_p(val_ptr), c_void_p(idx_ptr), MEM_DEVICE, c_void_p(stream or None)))

    def merge_shards_host(self, q, opts: QueryOpts, shard_val, shard_idx, nq=None, return_distance=True):
        q = _c_f64(q)
        shard_val = _c_f64(shard_val)
        shard_idx = np.ascontiguousarray(shard_idx, dtype=np.int64)
        n_shards = shard_val.shape[0]
        if q is not None:
            nq = q.shape[0]
        k = opts.n_neighbors
        idx = np.empty((nq, k), dtype=np.int64)
        dist = np.empty((nq, k), dtype=np.float64) if return_distance else None
        check(load().sknnr_merge_shards(self.handle, _host_ptr(q), nq, byref(opts), n_shards, _host_ptr(shard_val),
                                        _host_ptr(shard_idx), _host_ptr(dist), _host_ptr(idx), MEM_HOST, None))
        return dist, idx

    def merge_shards_device(self, q_ptr, nq, opts: QueryOpts, n_shards, val_ptr, sidx_ptr, dist_ptr, idx_ptr, stream=0):
        check(load().sknnr_merge_shards(self.handle, c_void_p(q_ptr or None), nq, byref(opts), n_shards, c_void_p(val_ptr),
                                        c_void_p(sidx_ptr), c_void_p(dist_ptr or None), c_void_p(idx_ptr), MEM_DEVICE,
                                        c_void_p(stream or None)))

    # ---- diagnostics ------------------------------------------------------------------------
    def debug_coarse_matrix(self, q):
        q = _c_f64(q)
        nq = q.shape[0]
        out = np.empty((nq, self.n_ref), dtype=np.float32)
        qn = np.empty(nq, dtype=np.float64)
        s, eps = c_double(), c_double()
        check(load().sknnr_debug_coarse_matrix(self.handle, _host_ptr(q), nq, _host_ptr(out),
                                               _host_ptr(qn), byref(s), byref(eps)))
        return out, qn, s.value, eps.value


class QueryStream:
    """Owner of one ``sknnr_stream*``: host tiles in, host results out, the PCIe pipeline kept full
    across pushes (see include/sknnr_hip.h).  Output arrays handed to :meth:`push` are filled at the
    latest when :meth:`flush` / :meth:`close` returns; the object keeps them alive until then."""

    def __init__(self, index: Index, opts: QueryOpts, want_dist=True, want_pred=False):
        self._index = index
        self._h = c_void_p()
        self._keep = []
        self.k = opts.n_neighbors
        self._opts = opts
        self.want_dist, self.want_pred = bool(want_dist), bool(want_pred)
        check(load().sknnr_stream_begin(index.handle, byref(opts), int(self.want_dist), int(self.want_pred),
                                        byref(self._h)))

    def push(self, q, out_idx=None, out_dist=None, out_pred=None, need_idx=True):
        """Answer the rows of ``q``; returns the (idx, dist, pred) arrays that will hold the results
        (the ones passed in, or fresh ones; ``need_idx=False`` with predictions skips the indices)."""
        q = _c_rows(q, self._opts)
        nq = q.shape[0]
        if out_idx is None and (need_idx or not self.want_pred):
            out_idx = np.empty((nq, self.k), dtype=np.int64)
        if out_dist is None and self.want_dist:
            out_dist = np.empty((nq, self.k), dtype=np.float64)
        if out_pred is None and self.want_pred:
            out_pred = np.empty((nq, self._index.t), dtype=np.float64)
        for a, dt, cols in ((out_idx, np.int64, self.k), (out_dist, np.float64, self.k),
                            (out_pred, np.float64, self._index.t)):
            if a is not None and (a.dtype != dt or not a.flags.c_contiguous or a.shape != (nq, cols)):
                raise ValueError(f"output arrays must be C-contiguous ({nq}, {cols}) {np.dtype(dt)}")
        check(load().sknnr_stream_push(self._h, _host_ptr(q), nq, _host_ptr(out_dist), _host_ptr(out_idx),
                                       _host_ptr(out_pred)))
        self._keep.append((out_idx, out_dist, out_pred))
        if len(self._keep) > 8:
            del self._keep[:-8]  # older tiles have left the pipeline (four slots: at most the last four pushes are pending)
        return out_idx, out_dist, out_pred

    def flush(self):
        check(load().sknnr_stream_flush(self._h))
        self._keep.clear()

    def close(self) -> int:
        """Flush and free; returns the number of rows pushed."""
        if not self._h:
            return 0
        n = c_int64(0)
        h, self._h = self._h, c_void_p()
        code = load().sknnr_stream_end(h, byref(n))
        self._keep.clear()
        check(code)
        return int(n.value)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()
        else:  # do not mask the original error
            try:
                self.close()
            except HipBackendError:
                pass

    def __del__(self):
        try:
            if self._h:
                load().sknnr_stream_end(self._h, None)
                self._h = c_void_p()
        except Exception:
            pass


def affine_transform_host(x, center=None, scale=None, proj=None, device: int = 0) -> np.ndarray:
    """``((x - center) / scale) @ proj`` on the GPU (float64 fma chains), host in / host out."""
    x = _c_f64(x)
    center, scale, proj = _c_f64(center), _c_f64(scale), _c_f64(proj)
    n, d_in = x.shape
    d = d_in if proj is None else proj.shape[1]
    out = np.empty((n, d), dtype=np.float64)
    check(load().sknnr_affine_transform(_host_ptr(x), n, d_in, _host_ptr(center), _host_ptr(scale),
                                        _host_ptr(proj), d, _host_ptr(out), device))
    return out


def crosswalk_host(table, idx, device: int = 0):
    table = np.ascontiguousarray(table, dtype=np.int64)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.empty(idx.shape, dtype=np.int64)
    check(load().sknnr_crosswalk(_host_ptr(table), table.size, _host_ptr(idx), idx.size,
                                 _host_ptr(out), device, MEM_HOST, None))
    return out


def crosswalk_device(table_ptr, n_table, idx_ptr, n, out_ptr, device=0, stream=0):
    check(load().sknnr_crosswalk(c_void_p(table_ptr), n_table, c_void_p(idx_ptr), n, c_void_p(out_ptr),
                                 device, MEM_DEVICE, c_void_p(stream or None)))
