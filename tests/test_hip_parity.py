"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs and against the committed golden vectors.

Bars (BASELINE.json north_star): neighbour indices bit-exact; distances and predictions
within 1e-5 relative of the reference.  Against the oracle the HIP path is in fact held to
bit-equality of float64 distances (same fma chains), which the tests assert."""

from __future__ import annotations

import numpy as np
import pytest

from conftest import assert_neighbors_match, load_golden, yaimpute_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from sknnr_amd import _native

    assert _native.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return _native


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


def _synth(n_ref, nq, d, **kw):
    from sknnr_amd import synth

    return synth.make_problem(n_ref, nq, d, t=kw.pop("t", 6), **kw)


# ---------------------------------------------------------------------------------------------
# the MFMA pre-filter itself: operand maps and error budget
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [3, 8, 16, 17, 32, 64, 100, 128])
def test_coarse_error_budget(N, d):
    """Every value of the split f16x3 contraction must be within the budget the certificate
    assumes; a wrong fragment map would show up here as O(1) errors, not as slow fallbacks."""
    x_ref, _, x_q = _synth(1500, 200, d, t=2)
    ix = N.Index(x_ref)
    m, qn, s, eps_c = ix.debug_coarse_matrix(x_q)
    mu = x_ref.mean(axis=0)
    rp, qp = s * (x_ref - mu), s * (x_q - mu)
    exact = (rp * rp).sum(1)[None, :] - 2.0 * qp @ rp.T
    np.testing.assert_allclose(qn, (qp * qp).sum(1), rtol=1e-12)
    unit = (np.sqrt((qp * qp).sum(1))[:, None] + np.sqrt((rp * rp).sum(1))[None, :]) ** 2
    ratio = np.abs(m - exact) / (eps_c * unit)
    assert ratio.max() < 0.5, f"coarse error uses {ratio.max():.2f} of the budget"
    ix.close()


@pytest.mark.parametrize("scale", [1e-6, 1e-3, 1.0, 1e3, 1e6])
def test_coarse_is_scale_invariant(N, scale):
    x_ref, _, x_q = _synth(600, 64, 24, t=2)
    ix = N.Index(x_ref * scale + 7.0 * scale)
    m, qn, s, eps_c = ix.debug_coarse_matrix(x_q * scale + 7.0 * scale)
    mu = (x_ref * scale + 7.0 * scale).mean(axis=0)
    rp, qp = s * (x_ref * scale + 7.0 * scale - mu), s * (x_q * scale + 7.0 * scale - mu)
    exact = (rp * rp).sum(1)[None, :] - 2.0 * qp @ rp.T
    unit = (np.sqrt((qp * qp).sum(1))[:, None] + np.sqrt((rp * rp).sum(1))[None, :]) ** 2
    assert (np.abs(m - exact) / (eps_c * unit)).max() < 0.5
    ix.close()


# ---------------------------------------------------------------------------------------------
# kneighbors / predict against the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [2, 8, 16, 17, 32, 64, 100])
@pytest.mark.parametrize("k", [1, 5, 7])
def test_kneighbors_matches_oracle(N, O, d, k):
    x_ref, y, x_q = _synth(2048, 1000, d)
    ix = N.Index(x_ref, y)
    for formula, fname in ((N.FORMULA_EXPANDED, "expanded"), (N.FORMULA_DIRECT, "direct")):
        dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k, formula=formula))
        od, oi = O.kneighbors(x_ref, x_q, k, fname)
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(dist, od)  # same fma chains -> bit-equal float64
        dist, idx = ix.kneighbors_host(None, ix.make_opts(k, formula=formula, exclude_self=True), nq=2048)
        od, oi = O.kneighbors(x_ref, None, k, fname)
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(dist, od)
    st = ix.stats()
    if k + 1 <= 7:
        assert st["coarse_queries"] == st["queries"]
        # (2-D: near ties are dense, ~1.3 % of the rows fall inside the analytic error bound)
        assert st["exact_fallbacks"] <= (0.02 if d <= 2 else 0.01) * st["queries"], st
    ix.close()


@pytest.mark.parametrize("deterministic", [True, False])
def test_exact_ties_and_zero_distances(N, O, deterministic):
    """Exact duplicate reference rows (distance ties) and queries that are copies of
    reference rows (zero / cancellation-noise distances)."""
    x_ref, y, x_q = _synth(2048, 1024, 32, n_dup_refs=96, n_dup_queries=64)
    ix = N.Index(x_ref, y)
    for k in (1, 5):
        dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k, deterministic=deterministic))
        od, oi = O.kneighbors(x_ref, x_q, k, "expanded", deterministic=deterministic)
        np.testing.assert_array_equal(dist, od)
        np.testing.assert_array_equal(idx, oi)  # tied rows too: the heap and quicksort are replayed
        dist, idx = ix.kneighbors_host(None, ix.make_opts(k, exclude_self=True, deterministic=deterministic), nq=2048)
        od, oi = O.kneighbors(x_ref, None, k, "expanded", deterministic=deterministic)
        np.testing.assert_array_equal(dist, od)
        np.testing.assert_array_equal(idx, oi)
    ix.close()


@pytest.mark.parametrize("deterministic", [True, False])
@pytest.mark.parametrize("k", [1, 3, 6])
def test_integer_features_many_exact_ties(N, O, deterministic, k):
    """Integer-valued features make exact float64 distance ties the norm; which tied row the
    reference keeps at the k-th slot depends on its heap's history, which the exact path
    replays.  Also the X=None corner case with more duplicates than neighbours."""
    rng = np.random.default_rng(7)
    x_ref = rng.integers(0, 4, size=(1500, 6)).astype(np.float64)
    x_q = rng.integers(0, 4, size=(400, 6)).astype(np.float64)
    y = rng.standard_normal((1500, 3))
    ix = N.Index(x_ref, y)
    # "direct" is the kd_tree stand-in: the oracle keeps tied rows in (d2, index) order there
    for formula, fname in ((N.FORMULA_EXPANDED, "expanded"), (N.FORMULA_DIRECT, "direct")):
        dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k, formula=formula, deterministic=deterministic))
        od, oi = O.kneighbors(x_ref, x_q, k, fname, deterministic=deterministic)
        np.testing.assert_array_equal(dist, od)
        np.testing.assert_array_equal(idx, oi)
        dist, idx = ix.kneighbors_host(None, ix.make_opts(k, formula=formula, exclude_self=True,
                                                          deterministic=deterministic), nq=1500)
        od, oi = O.kneighbors(x_ref, None, k, fname, deterministic=deterministic)
        np.testing.assert_array_equal(dist, od)
        np.testing.assert_array_equal(idx, oi)
    assert ix.stats()["exact_fallbacks"] > 0
    ix.close()


@pytest.mark.parametrize("k", [9, 14, 16, 30, 31, 40])
def test_larger_k(N, O, k):
    """k + self <= 31 stays on the MFMA path (first-generation kernel here: 700 reference rows; lists of 16 / 32 per
    lane -- the pooled lists of the second generation are covered by tests/test_round3_gpu.py); beyond that the whole
    call is answered by the exact scan."""
    x_ref, y, x_q = _synth(700, 300, 20)
    ix = N.Index(x_ref, y)
    dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k))
    od, oi = O.kneighbors(x_ref, x_q, k, "expanded")
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(dist, od)
    dist, idx = ix.kneighbors_host(None, ix.make_opts(k, exclude_self=True), nq=700)
    od, oi = O.kneighbors(x_ref, None, k, "expanded")
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(dist, od)
    st = ix.stats()
    if k == 40:
        assert st["exact_only_queries"] == 1000
    elif k + 1 <= 31:
        assert st["coarse_queries"] == 1000 and st["exact_fallbacks"] <= 20, st
    ix.close()


@pytest.mark.parametrize("d", [20, 40, 64, 72, 96, 112, 128])   # 16-wide K-steps 2..8: light / medium / heavy geometries
@pytest.mark.parametrize("k", [2, 6, 10, 20])                    # list lengths 6, 8, 16, 32
def test_every_launch_geometry(N, O, d, k):
    """Each (feature width, list length) pair selects its own workgroup shape, q-blocks per wave, LDS
    stage size and insertion code (coarse.hip.h, launch geometry): targets and the X=None self query
    must equal the oracle for all of them."""
    x_ref, y, x_q = _synth(2600, 1100, d, n_dup_refs=12, n_dup_queries=8)
    ix = N.Index(x_ref, y)
    dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k))
    od, oi = O.kneighbors(x_ref, x_q, k, "expanded")
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(dist, od)
    dist, idx = ix.kneighbors_host(None, ix.make_opts(k, exclude_self=True), nq=2600)
    od, oi = O.kneighbors(x_ref, None, k, "expanded")
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(dist, od)
    st = ix.stats()
    assert st["exact_only_queries"] == 0 and st["exact_fallbacks"] < 0.2 * st["queries"]
    ix.close()


@pytest.mark.parametrize("order", ["0", "1", "2", "3"])
def test_reference_image_order_does_not_change_results(N, O, monkeypatch, order):
    """The pre-filter sweeps the references in the caller's order or by increasing centred norm
    (chosen per index; SKNNR_IMAGE_ORDER pins it): candidates are mapped back to row indices, so the
    answers -- ties, X=None, row offsets included -- are the same either way."""
    monkeypatch.setenv("SKNNR_IMAGE_ORDER", order)
    x_ref, y, x_q = _synth(3100, 900, 24, n_dup_refs=40, n_dup_queries=20)
    ix = N.Index(x_ref, y)
    for k in (1, 5, 7):
        dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k, row_offset=1000))
        od, oi = O.kneighbors(x_ref, x_q, k, "expanded", row_offset=1000)
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(dist, od)
    dist, idx = ix.kneighbors_host(None, ix.make_opts(4, exclude_self=True), nq=3100)
    od, oi = O.kneighbors(x_ref, None, 4, "expanded")
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(dist, od)
    ix.close()


@pytest.mark.parametrize("d", [16, 64])
def test_larger_k_wider(N, O, d):
    x_ref, y, x_q = _synth(3000, 1500, d)
    ix = N.Index(x_ref, y)
    for k in (10, 20):
        dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k))
        od, oi = O.kneighbors(x_ref, x_q, k, "expanded")
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(dist, od)
    assert ix.stats()["exact_only_queries"] == 0
    ix.close()


def test_wide_features_use_the_exact_scan(N, O):
    x_ref, y, x_q = _synth(500, 200, 150)
    ix = N.Index(x_ref, y)
    dist, idx = ix.kneighbors_host(x_q, ix.make_opts(5))
    od, oi = O.kneighbors(x_ref, x_q, 5, "expanded")
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(dist, od)
    ix.close()


def test_certificate_failures_fall_back_to_the_exact_scan(N, O):
    """Force the rare branch: references packed so tightly (relative spread 1e-7) that the
    f32-class pre-filter cannot separate them -> certificates fail -> float64 scan; results
    must still be exact."""
    rng = np.random.default_rng(5)
    base = rng.standard_normal((1, 16)) * 100.0
    x_ref = base + 1e-5 * rng.standard_normal((3000, 16))
    x_ref[0] += 50.0  # one far row keeps the coarse scale large
    x_q = base + 1e-5 * rng.standard_normal((512, 16))
    ix = N.Index(x_ref)
    dist, idx = ix.kneighbors_host(x_q, ix.make_opts(5))
    od, oi = O.kneighbors(x_ref, x_q, 5, "expanded")
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(dist, od)
    st = ix.stats()
    assert st["exact_fallbacks"] > 0, "this input is meant to exercise the fallback"
    ix.close()


def test_overflowing_queries_are_still_exact(N, O):
    """Queries far outside the f16 range of the coarse image (|q| >> 512 x the reference
    spread) overflow to inf in the pre-filter and must be answered by the exact scan."""
    x_ref, y, x_q = _synth(1000, 64, 8)
    x_q[:8] *= 1e7
    ix = N.Index(x_ref)
    dist, idx = ix.kneighbors_host(x_q, ix.make_opts(3))
    od, oi = O.kneighbors(x_ref, x_q, 3, "expanded")
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_allclose(dist, od, rtol=1e-12)
    ix.close()


@pytest.mark.parametrize("n_ref", [1, 2, 7, 31, 33, 257])
def test_tiny_reference_sets(N, O, n_ref):
    x_ref, y, x_q = _synth(n_ref, 70, 5)
    ix = N.Index(x_ref, y)
    for k in sorted({1, min(3, n_ref), n_ref if n_ref <= 7 else 5}):
        dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k))
        od, oi = O.kneighbors(x_ref, x_q, k, "expanded")
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(dist, od)
    ix.close()


def test_errors_follow_the_reference(N):
    x_ref, y, x_q = _synth(10, 4, 3)
    ix = N.Index(x_ref, y)
    with pytest.raises(N.HipBackendError, match="Expected n_neighbors <= n_samples_fit, but n_neighbors = 11"):
        ix.kneighbors_host(x_q, ix.make_opts(11))
    with pytest.raises(N.HipBackendError, match="Expected n_neighbors < n_samples_fit, but n_neighbors = 10"):
        ix.kneighbors_host(None, ix.make_opts(10, exclude_self=True), nq=10)
    with pytest.raises(N.HipBackendError, match="Expected n_neighbors > 0"):
        ix.kneighbors_host(x_q, ix.make_opts(0))
    dist, idx = ix.kneighbors_host(x_q[:0], ix.make_opts(2))
    assert dist.shape == (0, 2) and idx.shape == (0, 2)
    ix.close()


@pytest.mark.parametrize("weights", ["uniform", "distance", yaimpute_weights])
@pytest.mark.parametrize("k", [1, 5, 7, 12])
def test_predict_matches_oracle(N, O, weights, k):
    from sknnr_amd._engine import KNNEngine

    x_ref, y, x_q = _synth(1500, 400, 16, t=9, n_dup_queries=16)
    eng = KNNEngine(x_ref, y)
    pred = eng.predict(x_q, k, weights)
    od, oi = O.kneighbors(x_ref, x_q, k, "expanded")
    np.testing.assert_allclose(pred, O.predict(y, od, oi, weights), rtol=1e-12, atol=0)
    pred = eng.predict(None, k, weights, exclude_self=True)
    od, oi = O.kneighbors(x_ref, None, k, "expanded")
    np.testing.assert_allclose(pred, O.predict(y, od, oi, weights), rtol=1e-12, atol=0)
    eng.close()


def test_affine_transform_and_fused_query_transform(N, O):
    from sknnr_amd import synth

    x_ref, y, x_q = _synth(900, 300, 12)
    rng = np.random.default_rng(3)
    center, scale = rng.standard_normal(12), 0.5 + rng.random(12)
    proj = rng.standard_normal((12, 20))
    for c, s, p in ((center, scale, proj), (center, None, proj), (center, scale, None), (None, None, None)):
        ref_t = N.affine_transform_host(x_ref, c, s, p)
        np.testing.assert_array_equal(ref_t, O.affine(x_ref, c, s, p))
        ix = N.Index(ref_t, y)
        ix.set_affine(12, c, s, p)
        dist, idx = ix.kneighbors_host(x_q, ix.make_opts(5, apply_affine=True))
        od, oi = O.kneighbors(ref_t, O.affine(x_q, c, s, p), 5, "expanded")
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(dist, od)
        ix.close()


def test_row_offset_and_chunk_independence(N, O):
    """A call split into shards that carry their global row offset gives the same answer as
    the whole call (the reorder's |idx - row| key)."""
    x_ref = np.array([1e-11, 1e-12, 1.0]).reshape(-1, 1)
    ix = N.Index(x_ref)
    q = np.zeros((6, 1))
    _, whole = ix.kneighbors_host(q, ix.make_opts(2))
    parts = [ix.kneighbors_host(q[a:b], ix.make_opts(2, row_offset=a))[1] for a, b in ((0, 1), (1, 4), (4, 6))]
    np.testing.assert_array_equal(np.vstack(parts), whole)
    np.testing.assert_array_equal(whole, O.kneighbors(x_ref, q, 2, "expanded")[1])
    assert whole[0].tolist() == [0, 1] and whole[1].tolist() == [1, 0]
    ix.close()


@pytest.mark.parametrize(("decimals", "expected"), [(8, [2, 1, 0]), (5, [1, 2, 0]), (2, [0, 1, 2])])
def test_precision_decimals(N, decimals, expected):
    x_ref = np.array([1e-3, 1e-6, 1e-9, 1.0]).reshape(-1, 1)
    ix = N.Index(x_ref)
    _, idx = ix.kneighbors_host(np.array([[0.0]]), ix.make_opts(3, decimals=decimals))
    assert idx[0].tolist() == expected
    ix.close()


def test_torch_device_tensors_round_trip(N, O):
    import torch

    from sknnr_amd._engine import KNNEngine

    x_ref, y, x_q = _synth(3000, 5000, 32, t=4)
    eng = KNNEngine(x_ref, y)
    xq = torch.as_tensor(x_q, device="cuda")
    dist, idx = eng.kneighbors(xq, 5)
    assert dist.is_cuda and idx.is_cuda and idx.dtype == torch.int64
    od, oi = O.kneighbors(x_ref, x_q, 5, "expanded")
    np.testing.assert_array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)
    pred = eng.predict(xq, 5, "distance")
    np.testing.assert_allclose(pred.cpu().numpy(), O.predict(y, od, oi, "distance"), rtol=1e-12, atol=0)
    ids = eng.crosswalk(idx, np.arange(3000, dtype=np.int64) + 100000)
    np.testing.assert_array_equal(ids.cpu().numpy(), oi + 100000)
    eng.close()


# ---------------------------------------------------------------------------------------------
# golden vectors from the reference, through the C ABI
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [8, 16, 32, 64])
@pytest.mark.parametrize("dup", [False, True])
def test_reference_goldens_synthetic(N, d, dup):
    from oracle import oracle as O

    g = load_golden(f"synth_raw_d{d}{'_dup' if dup else ''}.npz")
    x_ref, y, x_q = _synth(2048, 1024, d, n_dup_refs=96 if dup else 0, n_dup_queries=64 if dup else 0)
    ix = N.Index(x_ref, y)
    for k in (1, 5, 7):
        formula = N.FORMULA_EXPANDED if O.fit_method(2048, d, k) == "brute" else N.FORMULA_DIRECT
        dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k, formula=formula))
        assert_neighbors_match(idx, dist, g[f"tgt_k{k}_nn"], g[f"tgt_k{k}_dist"], fit_X=x_ref, atol=2e-6)
        dist, idx = ix.kneighbors_host(None, ix.make_opts(k, formula=formula, exclude_self=True), nq=2048)
        assert_neighbors_match(idx, dist, g[f"ref_k{k}_nn"], g[f"ref_k{k}_dist"], fit_X=x_ref, atol=2e-6)
        if not dup:
            pred = ix.predict_host(x_q, ix.make_opts(k, formula=formula, weight_mode=N.WEIGHTS_DISTANCE))
            np.testing.assert_allclose(pred, g[f"pred_distance_k{k}"], rtol=1e-5, atol=1e-8)
    ix.close()


def test_chunk_boundaries_and_row_offsets_in_one_big_call(N, O):
    """More query rows than one workspace chunk (4,194,304): rows around the chunk boundary, the
    last rows and rows whose tie-break depends on their global position must match the oracle."""
    import torch

    from sknnr_amd import synth
    from sknnr_amd._engine import KNNEngine

    n_ref, d, k = 3000, 8, 3
    nq = 4194304 + 70001
    x_ref = synth.make_features(n_ref, d, seed=0)
    x_ref[1] = x_ref[0]  # an exact duplicate pair: |idx - row| decides their order for nearby queries
    eng = KNNEngine(x_ref)
    g = torch.Generator(device="cuda").manual_seed(5)
    xq = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(
        synth.mixing_matrix(d), device="cuda")
    probe = [0, 1, 4194303, 4194304, 4194305, nq - 1]
    for r in probe:
        xq[r] = torch.as_tensor(x_ref[0], device="cuda")  # equidistant to rows 0 and 1
    dist, idx = eng.kneighbors(xq, k, row_offset=0)
    for a, b in ((0, 2000), (4194304 - 1000, 4194304 + 1000), (nq - 2000, nq)):
        od, oi = O.kneighbors(x_ref, xq[a:b].cpu().numpy(), k, "expanded", row_offset=a)
        np.testing.assert_array_equal(idx[a:b].cpu().numpy(), oi)
        np.testing.assert_array_equal(dist[a:b].cpu().numpy(), od)
    got = idx[probe, :2].cpu().numpy().tolist()
    assert got[0] == [0, 1] and got[1] == [1, 0]          # rows 0 and 1: |idx - row| picks the own index first
    assert all(p == [1, 0] for p in got[2:])              # far rows: 1 is closer to the row number
    eng.close()


def test_host_buffer_pipeline_matches_device_path(N, O):
    """numpy in / numpy out goes through the pinned two-slot pipeline (1,048,576 rows per slot):
    five slots' worth of rows, with and without distances / neighbours, must equal the
    device-resident call bit for bit, and the oracle around the slot boundaries."""
    import torch

    from sknnr_amd._engine import KNNEngine

    n_ref, d, t, k = 2500, 6, 3, 4
    slot = 1 << 20
    nq = 4 * slot + 12345
    x_ref, y, _ = _synth(n_ref, 10, d, t=t)
    rng = np.random.default_rng(11)
    xq = x_ref[rng.integers(0, n_ref, nq)] + 0.05 * rng.standard_normal((nq, d))
    ix = N.Index(x_ref, y)
    o = ix.make_opts(k, weight_mode=1)
    dist, idx = ix.kneighbors_host(xq, o)
    _, idx_only = ix.kneighbors_host(xq, o, return_distance=False)
    pred, dist_p, idx_p = ix.predict_host(xq, o, return_neighbors=True)
    pred_only = ix.predict_host(xq, o)
    np.testing.assert_array_equal(idx, idx_only)
    np.testing.assert_array_equal(idx, idx_p)
    np.testing.assert_array_equal(dist, dist_p)
    np.testing.assert_array_equal(pred, pred_only)
    eng = KNNEngine(x_ref, y)
    xd = torch.as_tensor(xq, device="cuda")
    dd, di = eng.kneighbors(xd, k)
    np.testing.assert_array_equal(di.cpu().numpy(), idx)
    np.testing.assert_array_equal(dd.cpu().numpy(), dist)
    pd_ = eng.predict(xd, k, weights="distance")
    np.testing.assert_array_equal(pd_.cpu().numpy(), pred)
    for a, b in ((0, 500), (slot - 300, slot + 300), (4 * slot - 300, 4 * slot + 300), (nq - 500, nq)):
        od, oi = O.kneighbors(x_ref, xq[a:b], k, "expanded", row_offset=a)
        np.testing.assert_array_equal(idx[a:b], oi)
        np.testing.assert_array_equal(dist[a:b], od)
        np.testing.assert_allclose(pred[a:b], O.predict(y, od, oi, "distance"), rtol=1e-12, atol=0)
    ix.close()
    eng.close()


def test_engine_writes_into_caller_tensors(N, O):
    """out=(dist, idx): results land in the caller's (slot of a larger) tensors; wrong shapes raise."""
    import torch

    from sknnr_amd._engine import KNNEngine

    x_ref, _, x_q = _synth(1500, 700, 10)
    eng = KNNEngine(x_ref)
    xq = torch.as_tensor(x_q, device="cuda")
    k = 4
    big_d = torch.full((3 * 700, k), -1.0, dtype=torch.float64, device="cuda")
    big_i = torch.full((3 * 700, k), -1, dtype=torch.int64, device="cuda")
    d, i = eng.kneighbors(xq, k, row_offset=700, out=(big_d[700:1400], big_i[700:1400]))
    od, oi = O.kneighbors(x_ref, x_q, k, "expanded", row_offset=700)
    np.testing.assert_array_equal(big_i[700:1400].cpu().numpy(), oi)
    np.testing.assert_array_equal(big_d[700:1400].cpu().numpy(), od)
    assert d.data_ptr() == big_d[700:1400].data_ptr() and i.data_ptr() == big_i[700:1400].data_ptr()
    assert (big_i[:700] == -1).all() and (big_i[1400:] == -1).all()   # neighbours' slots untouched
    with pytest.raises(ValueError):
        eng.kneighbors(xq, k, out=(big_d[:699], big_i[:699]))
    with pytest.raises(ValueError):
        eng.kneighbors(x_q, k, out=(big_d[:700], big_i[:700]))          # numpy input
    eng.close()


def test_two_indexes_alive_and_reused(N, O):
    """Handles are independent and reusable across calls of different shapes."""
    a_ref, a_y, a_q = _synth(900, 300, 12)
    b_ref, b_y, b_q = _synth(1700, 500, 40)
    ia, ib = N.Index(a_ref, a_y), N.Index(b_ref, b_y)
    for _ in range(2):
        for ix, ref, q in ((ia, a_ref, a_q), (ib, b_ref, b_q), (ia, a_ref, a_q[:7])):
            for k in (2, 5):
                dist, idx = ix.kneighbors_host(q, ix.make_opts(k))
                od, oi = O.kneighbors(ref, q, k, "expanded")
                np.testing.assert_array_equal(idx, oi)
                np.testing.assert_array_equal(dist, od)
    ia.close()
    ib.close()


# ---------------------------------------------------------------------------------------------
# BASELINE-size properties (no CPU reference at these sizes)
# ---------------------------------------------------------------------------------------------
def test_full_size_properties(N, O):
    """1M x 10k x 16 (BASELINE config 2 shape): sortedness, self-consistency under row
    permutation, and a seeded subsample checked against the oracle."""
    import torch

    from sknnr_amd import synth
    from sknnr_amd._engine import KNNEngine

    n_ref, nq, d, k = 10000, 1 << 20, 16, 5
    x_ref = synth.make_features(n_ref, d, seed=0)
    eng = KNNEngine(x_ref)
    g = torch.Generator(device="cuda").manual_seed(11)
    xq = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(
        synth.mixing_matrix(d), device="cuda")
    dist, idx = eng.kneighbors(xq, k, deterministic=False)
    assert bool((dist[:, 1:] >= dist[:, :-1]).all()), "distances must ascend"
    assert int(idx.min()) >= 0 and int(idx.max()) < n_ref
    assert bool((idx.sort(dim=1).values[:, 1:] != idx.sort(dim=1).values[:, :-1]).all()), "neighbours are distinct"
    # permutation invariance (without the position-dependent reorder)
    perm = torch.randperm(nq, device="cuda", generator=g)
    dist_p, idx_p = eng.kneighbors(xq[perm].contiguous(), k, deterministic=False)
    assert torch.equal(idx_p, idx[perm]) and torch.equal(dist_p, dist[perm])
    # subsample against the oracle
    sub = torch.arange(0, nq, 257, device="cuda")
    od, oi = O.kneighbors(x_ref, xq[sub].cpu().numpy(), k, "expanded", deterministic=False)
    np.testing.assert_array_equal(idx[sub].cpu().numpy(), oi)
    np.testing.assert_array_equal(dist[sub].cpu().numpy(), od)
    st = eng.stats()
    assert st["exact_fallbacks"] <= 1e-3 * st["queries"], st
    eng.close()
