"""Round-3 GPU tests: items of VERDICT r2 / ADVICE r2 that are not covered elsewhere."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from sknnr_amd import _native

    assert _native.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return _native


def test_callable_weights_over_tiles_carry_the_global_row(N):
    """``predict_chunks`` with a callable ``weights`` answers tile by tile; the reorder's
    second key |idx - row| (REF src/sknnr/_base.py:171) must use the row's position in the whole call, so that
    tiles give ``predict(np.concatenate(tiles))`` -- here reference rows 3 and 13 are identical and the weights
    depend on the neighbour's POSITION (first neighbour 2, second 1): rows 0-8 list [3, 13], rows 9-15 [13, 3]."""
    import sknnr_amd

    rng = np.random.default_rng(5)
    x_ref = rng.standard_normal((16, 20))
    x_ref[13] = x_ref[3]
    y = rng.standard_normal((16, 3))
    q = np.repeat(x_ref[3:4] + 1e-3, 16, axis=0)

    def positional(d):
        return np.tile(np.array([2.0, 1.0]), (d.shape[0], 1))

    est = sknnr_amd.RawKNNRegressor(n_neighbors=2, weights=positional).fit(x_ref, y)
    _, idx = est.kneighbors(q)
    assert idx[:9].tolist() == [[3, 13]] * 9 and idx[9:].tolist() == [[13, 3]] * 7, idx
    whole = est.predict(q)
    assert not np.array_equal(whole[0], whole[15])  # the order matters to this callable
    tiles = [q[:5], q[5:10], q[10:11], q[11:]]
    np.testing.assert_array_equal(est.predict_chunks(iter(tiles)), whole)
    out = np.empty((16, 3))
    np.testing.assert_array_equal(est.predict_chunks(iter(tiles), out=out), whole)


# ---------------------------------------------------------------------------------------------
# reference-sharded search (SURVEY 8e "alternative"; include/sknnr_hip.h: sknnr_shard_candidates / sknnr_merge_shards)
# ---------------------------------------------------------------------------------------------
def _sharded(N, x_ref, x_q, k, n_shards, formula, deterministic, self_rows=False, device_mem=False):
    """Candidates of `n_shards` shard handles, gathered as (shard, row, slot), merged on a handle over all rows."""
    from sknnr_amd.distributed import shard_bounds

    f = {"expanded": N.FORMULA_EXPANDED, "direct": N.FORMULA_DIRECT}[formula]
    kk = k + (1 if self_rows else 0)
    q = x_ref if self_rows else x_q
    vals, idxs = [], []
    for g in range(n_shards):
        a, b = shard_bounds(len(x_ref), n_shards, g)
        ix = N.Index(x_ref[a:b])
        v, i = ix.shard_candidates_host(q, ix.make_opts(kk, formula=f, deterministic=False), index_offset=a)
        assert (np.diff(v, axis=1) >= 0).all() and i.min() >= a and i.max() < b
        vals.append(v)
        idxs.append(i)
        ix.close()
    full = N.Index(x_ref)
    o = full.make_opts(k, formula=f, deterministic=deterministic, exclude_self=self_rows)
    sv, si = np.stack(vals), np.stack(idxs)
    if device_mem:
        import torch

        tv, ti = torch.as_tensor(sv, device="cuda"), torch.as_tensor(si, device="cuda")
        tq = None if self_rows else torch.as_tensor(x_q, device="cuda")
        dd = torch.empty((len(q), k), dtype=torch.float64, device="cuda")
        di = torch.empty((len(q), k), dtype=torch.int64, device="cuda")
        full.merge_shards_device(0 if tq is None else tq.data_ptr(), len(q), o, n_shards, tv.data_ptr(), ti.data_ptr(),
                                 dd.data_ptr(), di.data_ptr())
        torch.cuda.synchronize()
        dist, idx = dd.cpu().numpy(), di.cpu().numpy()
    else:
        dist, idx = full.merge_shards_host(None if self_rows else x_q, o, sv, si, nq=len(q))
    want = full.kneighbors_host(None if self_rows else x_q, o, nq=len(q))
    full.close()
    return (dist, idx), want, (sv, si)


@pytest.mark.parametrize("formula", ["expanded", "direct"])
@pytest.mark.parametrize("deterministic", [True, False])
def test_reference_sharded_search_with_duplicates_across_shards(N, formula, deterministic):
    """The gloo test's problem (tests/test_distributed_cpu.py: exact duplicates on both sides of the shard boundaries,
    queries that are copies of them) through the HIP entry points: bit-equal to the oracle's single heap over all rows
    and to the unsharded HIP call, for given rows and for the X=None path, 2 and 3 shards."""
    from oracle import oracle as O
    from test_distributed_cpu import _ref_sharded_problem

    x_ref, x_q = _ref_sharded_problem()
    for n_shards in (2, 3):
        for self_rows in (False, True):
            (dist, idx), (wd, wi), (sv, si) = _sharded(N, x_ref, x_q, 4, n_shards, formula, deterministic, self_rows)
            od, oi = O.kneighbors(x_ref, None if self_rows else x_q, 4, formula, deterministic=deterministic)
            np.testing.assert_array_equal(idx, oi)
            np.testing.assert_array_equal(dist, od)
            np.testing.assert_array_equal(idx, wi)
            np.testing.assert_array_equal(dist, wd)
            # ... and the oracle's restatement of the merge (which the gloo test uses) says the same
            md, mi, _ = O.merge_shards(x_ref, None if self_rows else x_q, sv, si, 4, formula, deterministic=deterministic)
            np.testing.assert_array_equal(mi, idx)
            np.testing.assert_array_equal(md, dist)


def test_reference_sharded_search_on_the_mfma_path(N):
    """Shards large enough for the pre-filter (3 x 6,000 rows x 32 features, a few rows duplicated across shards), 20,000
    query rows, device memory: candidates come from the MFMA path in raw mode (squared values by (value, index)),
    the merge equals the unsharded call and the oracle."""
    from oracle import oracle as O
    from sknnr_amd import synth

    x_ref, _, x_q = synth.make_problem(18_000, 20_000, 32, t=2, n_dup_queries=32)
    x_ref[7000:7040] = x_ref[100:140]
    x_ref[13000:13020] = x_ref[100:120]
    for k, self_rows in ((5, False), (3, True)):
        (dist, idx), (wd, wi), _ = _sharded(N, x_ref, x_q, k, 3, "expanded", True, self_rows, device_mem=True)
        np.testing.assert_array_equal(idx, wi)
        np.testing.assert_array_equal(dist, wd)
        od, oi = O.kneighbors(x_ref, None if self_rows else x_q, k, "expanded")
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(dist, od)


def test_ref_sharded_knn_class_single_rank(tmp_path):
    """RefShardedKNN over a fitted estimator with one gloo rank on the GPU box: the class's own engine plumbing (shard
    engine with the estimator's affine map, candidates, all-gather, merge, X=None) against estimator.kneighbors."""
    import socket

    import torch.distributed as dist

    import sknnr_amd
    from sknnr_amd import synth
    from sknnr_amd.distributed import RefShardedKNN

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        x_ref, y, x_q = synth.make_problem(5000, 3000, 16, t=4, kind="positive", n_dup_queries=16)
        est = sknnr_amd.GNNRegressor(n_neighbors=4).fit(x_ref, y)
        sh = RefShardedKNN(est)
        for det in (True, False):
            d, i = sh.kneighbors(x_q, use_deterministic_ordering=det)
            wd, wi = est.kneighbors(x_q, use_deterministic_ordering=det)
            np.testing.assert_array_equal(i, wi)
            np.testing.assert_array_equal(d, wd)
        d, i = sh.kneighbors(None)
        wd, wi = est.kneighbors()
        np.testing.assert_array_equal(i, wi)
        np.testing.assert_array_equal(d, wd)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("d", [8, 32, 48, 64])
def test_more_neighbours_than_a_list_holds_on_pooled_lists(N, d):
    """k + self = 6 .. 7 on lists of 6, 8 .. 15 on lists of 8, 16 .. 23 on lists of 12 (d = 8: 16 waves, d = 32: 12 waves), 24 .. 31
    on lists of 16 (second-generation pre-filter):
    thresholds of a rank beyond one list over the two lists of a query kept as one pool (coarse2.hip.h, pair_union_rank,
    coarse2_rank_extra).  A query with more of its neighbours in one lane's half of the rows than that lane's list holds
    (about one in forty at 16 of 22) goes through the hand-over between the lists.  8,000 reference rows (v2 needs 4,096),
    some duplicated (ties across the k-th slot), 6,000 query rows, X=None too.  48 and 64 features: the three- and four-K-step
    instances (12 waves per workgroup from lists of 8 at four K-steps on; round 4 -- the first-generation kernel before).
    """
    from oracle import oracle as O
    from sknnr_amd import synth

    x_ref, y, x_q = synth.make_problem(8_000, 6_000, d, t=2, n_dup_refs=24, n_dup_queries=16)
    ix = N.Index(x_ref, y)
    for k in (6, 7, 8, 10, 11, 13, 14, 15, 16, 20, 21, 23, 24, 25, 26, 30, 31):
        dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k))
        od, oi = O.kneighbors(x_ref, x_q, k, "expanded")
        np.testing.assert_array_equal(idx, oi, err_msg=f"k={k}")
        np.testing.assert_array_equal(dist, od, err_msg=f"k={k}")
        st = ix.stats()
        assert st["coarse_queries"] >= 6_000 and st["exact_only_queries"] == 0, st
    for k in (6, 7, 10, 15, 20, 30):  # + self: 7, 8, 11, 16, 21, 31
        dist, idx = ix.kneighbors_host(None, ix.make_opts(k, exclude_self=True), nq=8_000)
        od, oi = O.kneighbors(x_ref, None, k, "expanded")
        np.testing.assert_array_equal(idx, oi, err_msg=f"k={k}")
        np.testing.assert_array_equal(dist, od, err_msg=f"k={k}")
    # the certificate still carries most rows (no wholesale fall-back to the scan)
    for k in (7, 10, 20):
        ix.reset_stats()
        ix.kneighbors_host(x_q, ix.make_opts(k))
        st = ix.stats()
        assert st["exact_fallbacks"] <= 0.05 * 6_000, (k, st)
    ix.close()


@pytest.mark.parametrize("trees", [37, 512, 600, 1100])
def test_hamming_rescore_over_several_chunks_of_trees(N, trees):
    """The weighted-Hamming re-score flags differing trees 512 at a time (hamming.hip.h): one, two and three chunks, a
    tree count that is not a multiple of 8, few distinct ids (ties at the k-th distance are the norm), real-valued and
    equal weights; X=None too.  Bit-equal to the oracle's float64 arithmetic."""
    from oracle import oracle as O

    rng = np.random.default_rng(trees)
    ref = rng.integers(0, 6, (1500, trees)).astype(np.float64)
    src = rng.integers(0, 1500, 700)
    q = np.where(rng.random((700, trees)) < 0.7, ref[src], rng.integers(0, 6, (700, trees)).astype(np.float64))
    ix = N.Index(ref)
    for w in (rng.random(trees) + 0.01, np.full(trees, 1.0 / trees)):
        ix.set_hamming_weights(w)
        for k in (1, 5, 12):
            dist, idx = ix.kneighbors_host(q, ix.make_opts(k, formula=N.FORMULA_HAMMING))
            od, oi = O.kneighbors_hamming(ref, q, w, k)
            np.testing.assert_array_equal(idx, oi)
            np.testing.assert_array_equal(dist, od)
        dist, idx = ix.kneighbors_host(None, ix.make_opts(4, formula=N.FORMULA_HAMMING, exclude_self=True), nq=1500)
        od, oi = O.kneighbors_hamming(ref, None, w, 4)
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(dist, od)
    ix.close()
