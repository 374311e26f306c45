/*
 * sknnr_hip.h -- C ABI of the MI355X (gfx950) backend for sknnr's
 * kneighbors()/predict() hot path.
 *
 * Drop-in boundary: the reference has no FFI; everything its
 * estimators need from the engine is behind two calls on a fitted
 * sklearn.neighbors.KNeighborsRegressor,
 *     RawKNNRegressor.kneighbors  -> super().kneighbors(X, n_neighbors, return_distance=True)
 *                                    /root/reference/src/sknnr/_base.py:162-164
 *     (inherited) predict         -> /root/reference/src/sknnr/_base.py:39, :346-348
 * plus the transform that precedes them (_base.py:236-239) and the post-steps that
 * follow them (_base.py:166-180).  Each entry point below names the reference
 * interface it replaces.  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returns
 * SKNNR_OK (0) or a negative sknnr_status; sknnr_last_error() gives the message of
 * the calling thread's last failure.  All matrices are row-major (C order) float64,
 * indices are int64 -- exactly what the reference hands to / gets from scikit-learn.
 * The caller owns every buffer it passes; the handle owns its device copies.
 * One call at a time per handle: entry points serialise on a per-handle mutex, and a call that is handed a
 * different stream than the previous one first makes that stream wait (hipStreamWaitEvent) for the previous
 * call's last kernel, because both use the handle's workspace.
 */
#ifndef SKNNR_HIP_H
#define SKNNR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden; exactly these declarations are exported. */
#pragma GCC visibility push(default)

#define SKNNR_ABI_VERSION 5

typedef enum sknnr_status {
    SKNNR_OK = 0,
    SKNNR_ERR_INVALID = -1,     /* bad argument (message says which) */
    SKNNR_ERR_K_TOO_LARGE = -2, /* n_neighbors > n_samples_fit: SKL/neighbors/_base.py:848-859 */
    SKNNR_ERR_NO_TARGETS = -3,  /* predict without targets */
    SKNNR_ERR_UNSUPPORTED = -4, /* outside the envelope of the HIP kernels (no CPU fallback exists) */
    SKNNR_ERR_HIP = -5,         /* HIP runtime failure (message carries hipGetErrorString) */
    SKNNR_ERR_NO_DEVICE = -6,   /* no usable gfx950 device */
    SKNNR_ERR_NONFINITE = -7    /* query (or reference) rows contain NaN or infinity; the message is scikit-learn's
                                   ("Input X contains NaN." / "Input X contains infinity or a value too large
                                   for dtype('float64')."): SKL/utils/validation.py _assert_all_finite, reached
                                   from SKL/neighbors/_base.py:838-845 and REF transformers' transform() */
} sknnr_status;

/* Where the caller's query/output buffers live. */
typedef enum sknnr_memspace {
    SKNNR_MEM_HOST = 0,  /* ordinary host pointers; the library stages through PCIe */
    SKNNR_MEM_DEVICE = 1 /* device pointers on the handle's GPU (e.g. torch tensor.data_ptr()) */
} sknnr_memspace;

/* Which float64 expression the reference's engine evaluates for a pair
 * (SKL/neighbors/_base.py:620-648 picks the engine at fit time). */
typedef enum sknnr_formula {
    SKNNR_FORMULA_EXPANDED = 0, /* "brute"/ArgKmin: |x|^2 - 2 x.y + |y|^2, clamped at 0
                                   (SKL/metrics/_pairwise_distances_reduction/_argkmin.pyx.tp:492-502) */
    SKNNR_FORMULA_DIRECT = 1,   /* "kd_tree" (D <= 15): sum (x - y)^2
                                   (SKL/metrics/_dist_metrics.pxd.tp:39-57) */
    SKNNR_FORMULA_HAMMING = 2   /* RFNN / GBNN: weighted Hamming distance of tree node ids,
                                   sum_t w_t [a_t != b_t] / sum_t w_t, as scipy's cdist(metric="hamming", w=w)
                                   evaluates it on the float64 ids (sums in tree order), reached from
                                   REF _weighted_trees.py:53-59 (algorithm="brute", metric="hamming") and
                                   :139-140 (metric_params={"w": hamming_weights_}) through
                                   SKL/neighbors/_base.py:896-926 (pairwise_distances_chunked +
                                   _kneighbors_reduce_func).  No square root; rows tied exactly at the k-th
                                   distance are taken lowest index first.  The reference takes what numpy's
                                   argpartition takes; a caller that wants exactly that choice gets the tied
                                   rows' full distance rows from sknnr_hamming_distances and runs argpartition
                                   on them (the Python layer's hamming_tie_policy("numpy"); INTEGRATION.md,
                                   "Hamming ties").  Needs sknnr_index_set_hamming_weights. */
} sknnr_formula;

/* Element type of the query rows handed to kneighbors / predict / a stream (opts->query_dtype).  Rasters come as
 * float32 / int16 / uint8 ...; the reference widens them to float64 on the host before any arithmetic
 * (validate_data(..., dtype=FLOAT_DTYPES) then `X - env_center_` in float64: REF transformers/_cca_transformer.py:78-87,
 * _ccora_transformer.py:67-70; integers become float64 in every transformer).  Here the kernel that reads the rows
 * widens them -- exact for every type below, so the results are those of the float64 call bit for bit -- and the rows
 * cross PCIe at their own width. */
typedef enum sknnr_dtype {
    SKNNR_DTYPE_F64 = 0,
    SKNNR_DTYPE_F32 = 1,
    SKNNR_DTYPE_I16 = 2,
    SKNNR_DTYPE_U16 = 3,
    SKNNR_DTYPE_U8 = 4,
    SKNNR_DTYPE_I32 = 5
} sknnr_dtype;

typedef enum sknnr_weight_mode {
    SKNNR_WEIGHTS_UNIFORM = 0,  /* np.mean over the k neighbours  (SKL/neighbors/_regression.py:254-255) */
    SKNNR_WEIGHTS_DISTANCE = 1, /* 1/d, rows containing d == 0 become a 0/1 mask (SKL/neighbors/_base.py:113-119) */
    SKNNR_WEIGHTS_EXPLICIT = 2  /* caller supplies w (nq, k): result of a Python callable on the distances */
} sknnr_weight_mode;

typedef struct sknnr_index sknnr_index; /* opaque handle */

/* Per-call options of kneighbors/predict.  Zero-initialise, then set fields. */
typedef struct sknnr_query_opts {
    int32_t n_neighbors;   /* k of this call (reference: n_neighbors argument / ctor value) */
    int32_t exclude_self;  /* 1 = the X=None path: the query rows ARE reference rows
                              [row_offset, row_offset + nq); k+1 are searched and each row's own
                              index is dropped (SKL/neighbors/_base.py:828-833, :936-963) */
    int32_t deterministic; /* 1 = apply sknnr's tie-break reorder (REF _base.py:166-175) */
    int32_t decimals;      /* RawKNNRegressor.DISTANCE_PRECISION_DECIMALS (REF _base.py:102), default 10 */
    int32_t formula;       /* sknnr_formula */
    int32_t apply_affine;  /* 1 = queries are untransformed (d_in columns) and the handle's affine map
                              is applied first (REF _base.py:236-239); 0 = already transformed (d columns) */
    int32_t weight_mode;   /* predict only: sknnr_weight_mode */
    int32_t check_finite;  /* 1 = the kernels that read the query rows also test them for NaN / infinity
                              (what validate_data(ensure_all_finite=True) does on the host in the reference).
                              Host-memory calls then fail with SKNNR_ERR_NONFINITE; device-memory calls stay
                              asynchronous and the caller polls sknnr_check_finite() */
    int32_t query_dtype;   /* sknnr_dtype of the query rows `q` (0 = float64).  Other types need the MFMA envelope
                              (d <= 128) and a Euclidean formula; they are widened by the kernel that reads them */
    int32_t reserved_;     /* keep 0 */
    int64_t row_offset;    /* position of query row 0 inside the logical call: key 2 of the reorder is
                              |idx - row| with row counted over the whole call (REF _base.py:171), so a
                              shard or chunk must carry its global offset */
} sknnr_query_opts;

/* Counters of the handle since creation (or the last reset). */
typedef struct sknnr_stats {
    int64_t queries;           /* query rows answered */
    int64_t coarse_queries;    /* rows that went through the f16x3 MFMA pre-filter */
    int64_t exact_fallbacks;   /* rows whose certificate failed and were re-scanned in float64 */
    int64_t exact_only_queries;/* rows answered by the float64 scan alone (k or d outside the MFMA envelope) */
    double  last_kernel_ms;    /* device time of the most recent call (hipEvent, launch stream) */
    double  last_coarse_ms;    /* ... of which the MFMA pre-filter kernel */
    double  total_kernel_ms;   /* device time of all calls since the last reset (sums the chunks of a step) */
    double  total_coarse_ms;   /* ... of which the MFMA pre-filter kernel */
    int64_t timed_calls;       /* calls summed in the two totals */
    int64_t coarse_rows_timed; /* query rows processed by the pre-filter launches that total_coarse_ms sums: all rows of a
                                  call, except that when the thin last round runs beside the finaliser (side stream)
                                  only the 16-wave bulk launch is timed -- the rows to price its time against */
    double  mfma_executed_ratio; /* matrix work the most recent pre-filter launch ISSUED over the algorithmic
                                  2 nq n_ref d: reference rows padded to whole tiles / stages, the seed window swept
                                  twice, K padded to 16 (second-generation kernel: exact; first generation: its main
                                  products only, the correction products of visited tiles come on top) */
} sknnr_stats;

/* ---- lifetime -------------------------------------------------------------------------- */

/* Number of visible HIP devices (0 if none).  Never fails. */
int32_t sknnr_device_count(void);

/* ABI version of the loaded library (== SKNNR_ABI_VERSION). */
int32_t sknnr_abi_version(void);

/* Message of this thread's last error ("" if none). */
const char* sknnr_last_error(void);

/*
 * Build the device-resident index from the transformed reference rows.
 * Replaces KNeighborsRegressor.fit -> NeighborsBase._fit storing _fit_X and _y
 * (SKL/neighbors/_base.py:474-694; called from REF _base.py:107, :266-267).
 *   ref   : host, (n_ref, d) transformed features (the reference's _fit_X); every value finite, else
 *           SKNNR_ERR_NONFINITE (the reference's fit raises the same ValueError from its input validation)
 *   y     : host, (n_ref, t) targets (the reference's _y) or NULL (kneighbors only)
 *   device: HIP device ordinal
 */
int sknnr_index_create(const double* ref, int64_t n_ref, int32_t d, const double* y, int32_t t,
                       int32_t device, sknnr_index** out);

/* Free the handle and everything it owns.  NULL is allowed. */
void sknnr_index_destroy(sknnr_index* index);

/*
 * Install the query-time feature transform X -> ((X - center) / scale) @ proj.
 * Replaces transformer_.transform(X) (REF _base.py:236-239;
 * transformers/_cca_transformer.py:87, _ccora_transformer.py:70,
 * _mahalanobis_transformer.py:55, SKL/preprocessing/_data.py:1057-1098).
 *   center, scale : host, (d_in) or NULL;  proj : host, (d_in, d) or NULL (then d_in == d).
 */
int sknnr_index_set_affine(sknnr_index* index, int32_t d_in, const double* center,
                           const double* scale, const double* proj);

/*
 * Weights of the weighted-Hamming distance (one per column of the node-id matrix; finite, >= 0, not all 0).
 * Replaces metric_params={"w": self.hamming_weights_} (REF _weighted_trees.py:139-140).  The index then
 * holds node ids (float64 copies of the int64 ids the transformers emit, REF
 * transformers/_tree_node_transformer.py:177-201) and answers opts->formula = SKNNR_FORMULA_HAMMING.
 */
int sknnr_index_set_hamming_weights(sknnr_index* index, const double* w, int32_t n);

/*
 * The same affine map as a stand-alone call (no handle): out = ((x - center) / scale) @ proj.
 * Replaces X_transformed = self.transformer_.transform(X) at fit time (REF _base.py:251), so
 * that the stored reference rows and later query rows go through one and the same float64
 * fma chain.  x, out: host, (n, d_in) and (n, d); center/scale/proj as in set_affine.
 */
int sknnr_affine_transform(const double* x, int64_t n, int32_t d_in, const double* center,
                           const double* scale, const double* proj, int32_t d, double* out,
                           int32_t device);

/* Read back sizes: any pointer may be NULL. */
int sknnr_index_shape(const sknnr_index* index, int64_t* n_ref, int32_t* d, int32_t* t,
                      int32_t* d_in, int32_t* device);

int sknnr_get_stats(const sknnr_index* index, sknnr_stats* out);
int sknnr_reset_stats(sknnr_index* index);

/*
 * Poll (and clear) the non-finite-input flag raised by calls made with opts->check_finite = 1 on device
 * memory.  Synchronises `stream` (the stream those calls ran on).  Returns SKNNR_OK or
 * SKNNR_ERR_NONFINITE with scikit-learn's message.  Replaces the finiteness half of
 * validate_data(..., ensure_all_finite=True) (REF transformers/_cca_transformer.py:78-86,
 * SKL/neighbors/_base.py:838-845) for rows that never visit the host.
 */
int sknnr_check_finite(sknnr_index* index, void* stream);

/* ---- the hot path ---------------------------------------------------------------------- */

/*
 * k nearest reference rows of each query row.
 * Replaces RawKNNRegressor.kneighbors (REF _base.py:111-182) = sklearn's
 * KNeighborsMixin.kneighbors (SKL/neighbors/_base.py:763-963) + sknnr's reorder.
 *   q        : (nq, d_in or d) rows of opts->query_dtype (float64 unless set) in `mem`, or NULL with
 *              opts->exclude_self = 1 (the query rows are then the handle's own reference rows)
 *   out_dist : (nq, k) float64 in `mem`, ascending / reordered distances; may be NULL
 *   out_idx  : (nq, k) int64 in `mem`, reference row indices
 *   stream   : hipStream_t to launch on when mem == SKNNR_MEM_DEVICE (NULL = default stream);
 *              ignored for host buffers (the call then returns after the copy-back)
 */
int sknnr_kneighbors(sknnr_index* index, const void* q, int64_t nq,
                     const sknnr_query_opts* opts, double* out_dist, int64_t* out_idx,
                     int32_t mem, void* stream);

/*
 * Weighted multi-output mean of the neighbours' targets.
 * Replaces KNeighborsRegressor.predict (SKL/neighbors/_regression.py:224-268) as reached
 * from REF _base.py:39 (X=None, independent prediction) and :346-348.
 *   out_pred : (nq, t) float64 in `mem`
 *   out_dist, out_idx : optional (nq, k) outputs of the underlying kneighbors (NULL to skip)
 * With opts->weight_mode == SKNNR_WEIGHTS_EXPLICIT use sknnr_predict_from_neighbors instead.
 */
int sknnr_predict(sknnr_index* index, const void* q, int64_t nq, const sknnr_query_opts* opts,
                  double* out_pred, double* out_dist, int64_t* out_idx, int32_t mem, void* stream);

/*
 * The reduction alone, from neighbours already found (needed when `weights` is a Python
 * callable: the host evaluates it on the distances and passes w).
 *   dist : (nq, k) or NULL for uniform;  idx : (nq, k);  w : (nq, k) for SKNNR_WEIGHTS_EXPLICIT
 */
int sknnr_predict_from_neighbors(sknnr_index* index, const double* dist, const int64_t* idx,
                                 const double* w, int64_t nq, int32_t k, int32_t weight_mode,
                                 double* out_pred, int32_t mem, void* stream);

/*
 * Full weighted-Hamming distance rows: out[i, j] = distance between query row rows[i] and reference row j, in the
 * reference's float64 arithmetic (sknnr_formula, SKNNR_FORMULA_HAMMING) -- the matrix the reference's brute search
 * materialises chunk by chunk (SKL/neighbors/_base.py:896-926, pairwise_distances_chunked) before
 * _kneighbors_reduce_func (SKL/neighbors/_base.py:733-760) runs np.argpartition over each row.  For the rows whose
 * k-th distance is tied the caller can thus make numpy's own selection (REF tests/test_regressions.py:125-195 pin it).
 *   q    : (nq, d) float64 node ids in `mem`, or NULL: the handle's reference rows (the X=None path)
 *   rows : (n_rows) int64 in `mem`: which rows of q; NULL: rows 0 .. n_rows - 1
 *   out  : (n_rows, n_ref) float64 in `mem`
 */
int sknnr_hamming_distances(sknnr_index* index, const double* q, int64_t nq, const int64_t* rows, int64_t n_rows,
                            double* out, int32_t mem, void* stream);

/* ---- reference-sharded search ------------------------------------------------------------------ */

/*
 * When the REFERENCE rows are split over several handles (several GPUs), every shard answers every query row and the
 * per-shard answers are merged -- SURVEY.md section 8e, "alternative"; the reference's analogue is scikit-learn's
 * parallel-on-Y strategy: per-thread heaps over chunks of Y, then _parallel_on_Y_synchronize
 * (SKL/metrics/_pairwise_distances_reduction/_argkmin.pyx.tp:200-261), reached from REF _base.py:162-164.
 *
 * sknnr_shard_candidates: the n_neighbors nearest rows of THIS handle's reference rows (a shard) as raw candidates:
 *   out_val (nq, n_neighbors) the formula's values -- squared distances (expanded / direct), the Hamming distance --
 *   ascending by (value, index); out_idx the shard's row indices + index_offset (the shard's first row in the whole
 *   reference set).  No self exclusion (exclude_self must be 0: for the X=None path ask for n_neighbors + 1 and let
 *   the merge drop the row itself), no square root, no reorder.  The shard must hold at least n_neighbors rows.
 *
 * sknnr_merge_shards: the ranks' candidate arrays, gathered -- shard_val / shard_idx (n_shards, nq, kk), kk =
 *   n_neighbors + exclude_self, shard g's block at [g] -- merged into the call's final answer exactly as
 *   sknnr_kneighbors gives it (smallest (value, index) first, X=None drop, square root, sknnr's reorder).  `index` is a
 *   handle over ALL reference rows (every rank holds the small float64 copy; the shards split the sweep): a row whose
 *   merged answer is not unique -- an exact tie across the last slot, which the reference's heap settles by its
 *   history -- is re-scanned in float64 over all rows on this handle, the same replay of the reference's engine that
 *   sknnr_kneighbors uses for tied rows, so that a sharded call returns what the unsharded call returns.
 *   q / opts as for sknnr_kneighbors (q NULL with exclude_self = 1: the rows are reference rows [row_offset, +nq)).
 */
int sknnr_shard_candidates(sknnr_index* index, const double* q, int64_t nq, const sknnr_query_opts* opts,
                           int64_t index_offset, double* out_val, int64_t* out_idx, int32_t mem, void* stream);
int sknnr_merge_shards(sknnr_index* index, const double* q, int64_t nq, const sknnr_query_opts* opts, int32_t n_shards,
                       const double* shard_val, const int64_t* shard_idx, double* out_dist, int64_t* out_idx,
                       int32_t mem, void* stream);

/* ---- streamed query tiles (raster ingestion) ------------------------------------------------ */

/*
 * Wall-to-wall mapping feeds the hot path tile by tile: the reference's documented workflow predicts
 * plot IDs / attributes for every pixel of a raster (REF docs/pages/usage.md:101-128, README.md:66-67),
 * i.e. calls kneighbors(X_tile, return_dataframe_index=True) / predict(X_tile) once per block of pixels.
 * A stream keeps the handle's three-stream PCIe pipeline (copy-in | kernels | copy-out) full ACROSS
 * those calls and carries the global row offset itself, so that N pushes give bit for bit what one call
 * on the concatenated rows gives (key 2 of the reorder, REF _base.py:171, counts rows over the whole
 * logical call).
 *
 *   begin : opts as for sknnr_kneighbors / sknnr_predict (exclude_self must be 0); opts->row_offset is
 *           the global row of the first pushed row.  want_dist / want_pred say which optional outputs
 *           later pushes may ask for.  One open stream per handle; host-memory kneighbors/predict
 *           calls on the handle fail while it is open (device-memory calls are allowed).
 *   push  : q is a HOST (nq, d_in or d) tile of the stream's opts->query_dtype and may be reused as soon as the call returns.  The
 *           tile's results are written to the HOST buffers passed with it -- out_idx (nq, k), out_dist
 *           (nq, k) or NULL, out_pred (nq, t) or NULL -- at the latest when a flush or end returns (earlier in
 *           practice: a tile leaves its pipeline slot when the slot is needed again, four tiles later);
 *           the buffers must stay valid until then.
 *   flush : every pushed tile's results are in place on return.  With opts->check_finite the status
 *           is SKNNR_ERR_NONFINITE if any pushed value was NaN or infinite.
 *   end   : flush, then free the stream (NULL is allowed); *rows_pushed (optional) = total rows.
 * A stream refers to its index: end it before sknnr_index_destroy.
 */
typedef struct sknnr_stream sknnr_stream;
int sknnr_stream_begin(sknnr_index* index, const sknnr_query_opts* opts, int32_t want_dist, int32_t want_pred,
                       sknnr_stream** out);
int sknnr_stream_push(sknnr_stream* stream, const void* q, int64_t nq, double* out_dist, int64_t* out_idx,
                      double* out_pred);
int sknnr_stream_flush(sknnr_stream* stream);
int sknnr_stream_end(sknnr_stream* stream, int64_t* rows_pushed);

/*
 * Dataframe-index crosswalk: out[i] = table[idx[i]].
 * Replaces self.dataframe_index_in_[neigh_ind] (REF _base.py:177-180) for int64 plot IDs.
 *   table : (n_table) int64 in `mem`;  idx, out : (n) int64 in `mem`
 */
int sknnr_crosswalk(const int64_t* table, int64_t n_table, const int64_t* idx, int64_t n,
                    int64_t* out, int32_t device, int32_t mem, void* stream);

/* ---- diagnostics (used by the parity tests to validate the MFMA operand maps) ----------- */

/*
 * Full matrix of the pre-filter's approximate ranking values for a small problem:
 *   out[i, j] ~= s^2 (|r_j - mu|^2 - 2 (q_i - mu).(r_j - mu))   float32, host (nq, n_ref)
 * computed by the MFMA sequence of the first-generation kernel (all three split products on the matrix pipe;
 * the second-generation kernel's values differ by its smaller rounding budget only and are covered end to end).  Also returns the scale s,
 * the per-query |s (q_i - mu)|^2 (host, nq) and the error budget eps the certificate uses.
 * q is host, already transformed (d columns).  nq * n_ref must be <= 2^24.
 */
int sknnr_debug_coarse_matrix(sknnr_index* index, const double* q, int64_t nq, float* out,
                              double* out_qnorm, double* out_scale, double* out_eps);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SKNNR_HIP_H */
