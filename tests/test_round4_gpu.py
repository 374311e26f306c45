"""Round-4 GPU tests: narrow query element types through the C ABI and what else round 4 adds that
the older files do not cover."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NARROW = [np.float32, np.int16, np.uint16, np.uint8, np.int32]


@pytest.fixture(scope="module")
def N():
    from sknnr_amd import _native

    assert _native.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return _native


def _rows(rng, n, d, dtype):
    """Rows that a raster of that element type could hold (integers: small counts; float32: reals)."""
    if np.issubdtype(dtype, np.floating):
        return rng.standard_normal((n, d)).astype(dtype)
    hi = {np.uint8: 200, np.int16: 3000, np.uint16: 6000, np.int32: 70000}[dtype]
    lo = -hi // 2 if np.issubdtype(dtype, np.signedinteger) else 0
    return rng.integers(lo, hi, (n, d)).astype(dtype)


@pytest.mark.parametrize("dtype", NARROW)
def test_narrow_rows_equal_the_float64_call_bit_for_bit(N, dtype):
    """sknnr_query_opts.query_dtype: float32 / int16 / uint16 / uint8 / int32 rows are widened by the kernel that reads
    them -- exactly, as the reference's host conversion does (validate_data then float64 arithmetic, REF
    transformers/_cca_transformer.py:78-87) -- so indices and float64 distances equal the float64 call's, with and
    without the affine map, host memory (odd and even column counts: the scalar and the 16-byte load paths) and
    device memory, kneighbors and predict."""
    import torch

    rng = np.random.default_rng(11)
    for d_in, d in ((9, 7), (12, 12), (32, 20)):
        ref_raw = _rows(rng, 3000, d_in, dtype)
        q = _rows(rng, 5000, d_in, dtype)
        center = ref_raw.astype(np.float64).mean(axis=0)
        scale = ref_raw.astype(np.float64).std(axis=0) + 0.5
        proj = rng.standard_normal((d_in, d)) / np.sqrt(d_in)
        ref_t = N.affine_transform_host(ref_raw.astype(np.float64), center, scale, proj)
        y = rng.standard_normal((3000, 3))
        ix = N.Index(ref_t, y)
        ix.set_affine(d_in, center, scale, proj)
        code = N.dtype_code(dtype)
        assert code and code > 0
        q64 = q.astype(np.float64)
        for k, weight_mode in ((5, N.WEIGHTS_UNIFORM), (1, N.WEIGHTS_DISTANCE)):
            o64 = ix.make_opts(k, apply_affine=True, check_finite=True, weight_mode=weight_mode)
            on = ix.make_opts(k, apply_affine=True, check_finite=True, weight_mode=weight_mode, query_dtype=code)
            d0, i0 = ix.kneighbors_host(q64, o64)
            d1, i1 = ix.kneighbors_host(q, on)
            np.testing.assert_array_equal(i1, i0)
            np.testing.assert_array_equal(d1, d0)
            np.testing.assert_array_equal(ix.predict_host(q, on), ix.predict_host(q64, o64))
            # device memory
            qd = torch.as_tensor(q, device="cuda")
            dd = torch.empty((len(q), k), dtype=torch.float64, device="cuda")
            di = torch.empty((len(q), k), dtype=torch.int64, device="cuda")
            ix.kneighbors_device(qd.data_ptr(), len(q), on, dd.data_ptr(), di.data_ptr())
            torch.cuda.synchronize()
            np.testing.assert_array_equal(di.cpu().numpy(), i0)
            np.testing.assert_array_equal(dd.cpu().numpy(), d0)
        ix.close()
        # no affine map (RawKNNRegressor on narrow features): the rows ARE the transformed rows
        ref2 = _rows(rng, 2500, d, dtype)
        q2 = _rows(rng, 3000, d, dtype)
        ix = N.Index(ref2.astype(np.float64))
        d0, i0 = ix.kneighbors_host(q2.astype(np.float64), ix.make_opts(4))
        d1, i1 = ix.kneighbors_host(q2, ix.make_opts(4, query_dtype=code))
        np.testing.assert_array_equal(i1, i0)
        np.testing.assert_array_equal(d1, d0)
        ix.close()


def test_narrow_rows_outside_the_envelope_are_refused_by_the_library_and_widened_by_python(N):
    import sknnr_amd

    rng = np.random.default_rng(2)
    ref = rng.standard_normal((400, 150))
    q = rng.standard_normal((50, 150)).astype(np.float32)
    ix = N.Index(ref)
    with pytest.raises(N.HipBackendError, match="needs d <= 128"):
        ix.kneighbors_host(q, ix.make_opts(3, query_dtype=N.dtype_code(np.float32)))
    with pytest.raises(ValueError, match="unknown query_dtype"):  # (the binding refuses before the library would)
        ix.kneighbors_host(q.astype(np.float64), ix.make_opts(3, query_dtype=17))
    import ctypes

    o = ix.make_opts(3, query_dtype=17)
    q64 = np.ascontiguousarray(q, dtype=np.float64)
    out_i = np.empty((50, 3), dtype=np.int64)
    rc = N.load().sknnr_kneighbors(ix.handle, q64.ctypes.data_as(ctypes.c_void_p), 50, ctypes.byref(o), None,
                                   out_i.ctypes.data_as(ctypes.c_void_p), N.MEM_HOST, None)
    assert rc == N.ERR_INVALID and b"unknown query_dtype" in N.load().sknnr_last_error()
    ix.close()
    est = sknnr_amd.RawKNNRegressor(n_neighbors=3).fit(ref, ref[:, :2])  # d > 128: exact scan only, float64 rows
    d0, i0 = est.kneighbors(q.astype(np.float64))
    d1, i1 = est.kneighbors(q)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.uint8])
def test_narrow_tiles_through_the_estimators_equal_the_float64_call(dtype):
    """VERDICT r3 item 5, 'Done': float32 / int16 (/ uint8) tiles through kneighbors_chunks / predict_chunks and the
    one-shot calls of a transformed estimator equal the float64 call bit for bit; the validation layer no longer widens
    them on the host (the arrays reach the engine in their own element type)."""
    import sknnr_amd
    from sknnr_amd import synth

    rng = np.random.default_rng(4)
    x_ref, y, _ = synth.make_problem(4000, 10, 24, t=6, kind="positive")
    if np.issubdtype(dtype, np.floating):
        q = rng.standard_normal((20_000, 24)).astype(dtype)
    else:
        q = rng.integers(0, 5, (20_000, 24)).astype(dtype)
    q64 = q.astype(np.float64)
    est = sknnr_amd.GNNRegressor(n_neighbors=5, weights="distance").fit(x_ref, y)
    seen = []
    real = est.regressor_.engine_._index.kneighbors_host

    def spy(qq, opts, **kw):
        seen.append((None if qq is None else qq.dtype, opts.query_dtype))
        return real(qq, opts, **kw)

    est.regressor_.engine_._index.kneighbors_host = spy
    d0, i0 = est.kneighbors(q64)
    d1, i1 = est.kneighbors(q)
    del est.regressor_.engine_._index.kneighbors_host
    assert seen[-1] == (np.dtype(dtype), sknnr_amd._native.dtype_code(dtype)) and seen[-2] == (np.dtype(np.float64), 0)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    np.testing.assert_array_equal(est.predict(q), est.predict(q64))
    tiles = [q[:7000], q[7000:7001], q[7001:15000], q[15000:]]
    d2, i2 = est.kneighbors_chunks(iter(tiles))
    np.testing.assert_array_equal(i2, i0)
    np.testing.assert_array_equal(d2, d0)
    np.testing.assert_array_equal(est.predict_chunks(iter(tiles)), est.predict(q64))
    out_d, out_i = np.empty((len(q), 5)), np.empty((len(q), 5), dtype=np.int64)
    est.kneighbors_chunks(iter(tiles), out=(out_d, out_i))
    np.testing.assert_array_equal(out_i, i0)
    # tiles of one streamed call share an element type; a float64 stream widens what comes later
    with pytest.raises(ValueError, match="share an element type"):
        est.kneighbors_chunks(iter([q[:100], q64[100:200]]))
    d3, i3 = est.kneighbors_chunks(iter([q64[:100], q[100:300]]))
    np.testing.assert_array_equal(i3, i0[:300])
    # non-finite float32 input is reported with scikit-learn's sentence, by the kernel that reads the rows
    if np.issubdtype(dtype, np.floating):
        bad = q[:50].copy()
        bad[3, 2] = np.nan
        with pytest.raises(ValueError, match="Input X contains NaN"):
            est.kneighbors(bad)
        bad[3, 2] = np.inf
        with pytest.raises(ValueError, match="infinity"):
            est.kneighbors_chunks(iter([bad]))
    # CUDA tensors keep their element type too
    import torch

    dt, it = est.kneighbors(torch.as_tensor(q, device="cuda"))
    np.testing.assert_array_equal(it.cpu().numpy(), i0)
    np.testing.assert_array_equal(dt.cpu().numpy(), d0)


def test_hamming_candidate_lists_are_compacted_before_they_overflow(N):
    """candidates admitted early against the loose running bound used to fill the 192 slots of a query
    for good (about kk (1 + ln(n_ref / kk)) admissions, plus ties: kk >= 16 at a million rows, sooner with uniform weights)
    and the query fell to the full float64 scan.  The list is now compacted against the current bound when it is full and
    against the final bound before it goes out: kk = 32 over 200,000 rows with uniform weights stays on the integer path,
    bit-equal to the oracle."""
    from oracle import oracle as O
    from sknnr_amd import synth

    ids_ref, ids_q = synth.make_forest_ids(200_000, 512, 60, seed=3)
    w = np.random.default_rng(9).random(60) + 0.05
    ix = N.Index(ids_ref)
    ix.set_hamming_weights(w)
    ix.reset_stats()
    d, i = ix.kneighbors_host(ids_q, ix.make_opts(32, formula=N.FORMULA_HAMMING))
    st = ix.stats()
    assert st["exact_fallbacks"] <= 0.05 * len(ids_q), st
    od, oi = O.kneighbors_hamming(ids_ref, ids_q[:64], w, 32)
    np.testing.assert_array_equal(i[:64], oi)
    np.testing.assert_array_equal(d[:64], od)
    ix.close()


def test_hamming_distance_rows_in_device_memory(N):
    """sknnr_hamming_distances with device pointers (rows selected by a device index list, results left on the device)."""
    import ctypes

    import torch
    from scipy.spatial.distance import cdist

    rng = np.random.default_rng(6)
    ref = rng.integers(0, 5, (300, 21)).astype(np.float64)
    q = rng.integers(0, 5, (40, 21)).astype(np.float64)
    w = rng.random(21) + 0.1
    ix = N.Index(ref)
    ix.set_hamming_weights(w)
    rows = torch.tensor([3, 39, 0, 3], dtype=torch.int64, device="cuda")
    qd = torch.as_tensor(q, device="cuda")
    out = torch.empty((4, 300), dtype=torch.float64, device="cuda")
    vp = ctypes.c_void_p
    N.check(N.load().sknnr_hamming_distances(ix.handle, vp(qd.data_ptr()), 40, vp(rows.data_ptr()), 4, vp(out.data_ptr()), N.MEM_DEVICE,
                                             vp(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), cdist(q[[3, 39, 0, 3]], ref, "hamming", w=w))
    with pytest.raises(N.HipBackendError, match="set_hamming_weights"):
        N.Index(ref).hamming_distances_host(q)
    ix.close()


@pytest.mark.parametrize("weights", ["uniform", "distance"])
def test_numpy_tie_policy_through_raw_regressor_cuda_tensors_and_predict(weights):
    """hamming_tie_policy("numpy") on tie-saturated node ids (three ids, integer weights): kneighbors of numpy rows and of
    CUDA tensors, predict with uniform / inverse-distance weights and the X=None path equal scikit-learn's brute Hamming
    regressor followed by the reference's reorder (REF _base.py:166-175) -- the reference's own pipeline."""
    import torch
    from sklearn.neighbors import KNeighborsRegressor

    import sknnr_amd

    rng = np.random.default_rng(12)
    ref = rng.integers(0, 3, (400, 12)).astype(np.float64)
    q = rng.integers(0, 3, (150, 12)).astype(np.float64)
    y = rng.standard_normal((400, 3))
    w = rng.integers(1, 4, 12).astype(np.float64)

    def reference(X, k=4):
        skl = KNeighborsRegressor(n_neighbors=k, algorithm="brute", metric="hamming", metric_params={"w": w}, weights=weights).fit(ref, y)
        d, i = skl.kneighbors(X)
        rounded = np.round(d / np.maximum(d.max(axis=1, keepdims=True), 1.0), decimals=10)
        order = np.lexsort((i, np.abs(i - np.arange(len(i))[:, None]), rounded), axis=1)
        d, i = np.take_along_axis(d, order, axis=1), np.take_along_axis(i, order, axis=1)
        # the reference's predict = sklearn's predict over ITS kneighbors (the reordered ones)
        if weights == "uniform":
            pred = y[i].mean(axis=1)
        else:
            with np.errstate(divide="ignore"):
                wt = 1.0 / d
            inf = np.isinf(wt)
            rows_inf = inf.any(axis=1)
            wt[rows_inf] = inf[rows_inf]
            pred = (y[i] * wt[:, :, None]).sum(axis=1) / wt.sum(axis=1)[:, None]
        return d, i, pred

    with sknnr_amd.hamming_tie_policy("numpy"):
        est = sknnr_amd.RawKNNRegressor(n_neighbors=4, algorithm="brute", metric="hamming", metric_params={"w": w}, weights=weights)
        est.fit(ref, y)
        rd, ri, rp = reference(q)
        d, i = est.kneighbors(q)
        np.testing.assert_array_equal(i, ri)
        np.testing.assert_array_equal(d, rd)
        assert est.regressor_._last_numpy_tie_rows > 50 if hasattr(est, "regressor_") else est._last_numpy_tie_rows > 50
        dt, it = est.kneighbors(torch.as_tensor(q, device="cuda"))
        np.testing.assert_array_equal(it.cpu().numpy(), ri)
        np.testing.assert_allclose(est.predict(q), rp, rtol=1e-12, atol=1e-12)
        sd, si, sp = reference(None)
        d, i = est.kneighbors()
        np.testing.assert_array_equal(i, si)
        np.testing.assert_allclose(est.independent_prediction_, sp, rtol=1e-12, atol=1e-12)
    # the default policy differs on these inputs (lowest index first) -- the mode is what makes them equal
    d0, i0 = est.kneighbors(q)
    assert not np.array_equal(i0, ri)
