"""The N>1 path on CPU: two gloo ranks shard the query rows, each answers its block with the
oracle as the local engine (tests may use the oracle; the product path uses the HIP engine
through the same ShardedKNN), results are all-gathered and must equal the single-call answer
-- including the reorder's global row offset and uneven blocks."""

from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_q, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from oracle import oracle as O
    from sknnr_amd import synth
    from sknnr_amd.distributed import ShardedKNN, shard_bounds

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x_ref, y, x_q = synth.make_problem(300, n_q, 6, t=3, n_dup_refs=20, n_dup_queries=10)
        x_q[1] = x_q[0]  # identical queries at different rows: the |idx - row| key matters

        def local_kn(block, row_offset, k, use_deterministic_ordering=True, n_self_rows=None):
            if block is None:
                d, i = O.argkmin(x_ref[row_offset:row_offset + n_self_rows], x_ref, k + 1, "expanded")
                d, i = O.drop_self(d, i, row_offset)
                return O.deterministic_reorder(d, i, 10, row_offset) if use_deterministic_ordering else (d, i)
            return O.kneighbors(x_ref, block, k, "expanded", deterministic=use_deterministic_ordering,
                                row_offset=row_offset)

        def local_pred(block, row_offset, n_self_rows=None):
            d, i = local_kn(block, row_offset, 4, n_self_rows=n_self_rows)
            return O.predict(y, d, i, "distance")

        sh = ShardedKNN(None, local_kneighbors=local_kn, local_predict=local_pred)
        d_all, i_all = sh.kneighbors(x_q, 4)
        a, b = shard_bounds(n_q, world, rank)
        d_blk, i_blk = sh.kneighbors(x_q[a:b], 4, n_rows_total=n_q, X_is_local_block=True)
        d_self, i_self = sh.kneighbors(None, 4, n_rows_total=300)
        p_all = sh.predict(x_q)
        p_self = sh.predict(None, n_rows_total=300)
        # chunk-cyclic dealing with in-place all-gathers (bench.py's N > 1 layout): 30 rows per rank
        # in chunks of 8 (ragged last chunk); rank r's chunk [a, b) holds global rows W*a + r*(b-a) ...
        import torch

        from sknnr_amd.distributed import cyclic_slot

        n_loc, chunk = 30, 8
        glob = np.concatenate([x_q, x_q[::-1]])[: world * n_loc] if 2 * n_q >= world * n_loc else None
        d_cyc = i_cyc = np.zeros((0, 4))
        if glob is not None:
            mine = np.concatenate([glob[cyclic_slot(world, rank, a, min(n_loc, a + chunk)):
                                        cyclic_slot(world, rank, a, min(n_loc, a + chunk)) + min(n_loc, a + chunk) - a]
                                   for a in range(0, n_loc, chunk)])
            d_c, i_c = sh.kneighbors_cyclic(torch.as_tensor(mine), 4, chunk_rows=chunk)
            d_cyc, i_cyc = d_c.numpy(), i_c.numpy()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), d_all=d_all, i_all=i_all, d_blk=d_blk, i_blk=i_blk,
                 d_self=d_self, i_self=i_self, p_all=p_all, p_self=p_self, d_cyc=d_cyc, i_cyc=i_cyc)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_q", [64, 37])
def test_two_rank_sharding_matches_single_call(tmp_path, n_q):
    import torch.multiprocessing as mp

    from oracle import oracle as O
    from sknnr_amd import synth

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_q, str(tmp_path)), nprocs=world, join=True)

    x_ref, y, x_q = synth.make_problem(300, n_q, 6, t=3, n_dup_refs=20, n_dup_queries=10)
    x_q[1] = x_q[0]
    d, i = O.kneighbors(x_ref, x_q, 4, "expanded")
    ds, is_ = O.kneighbors(x_ref, None, 4, "expanded")
    for rank in range(world):
        r = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        np.testing.assert_array_equal(r["i_all"], i)
        np.testing.assert_array_equal(r["d_all"], d)
        np.testing.assert_array_equal(r["i_blk"], i)
        np.testing.assert_array_equal(r["d_blk"], d)
        np.testing.assert_array_equal(r["i_self"], is_)
        np.testing.assert_array_equal(r["d_self"], ds)
        np.testing.assert_array_equal(r["p_all"], O.predict(y, d, i, "distance"))
        np.testing.assert_array_equal(r["p_self"], O.predict(y, ds, is_, "distance"))
        if 2 * n_q >= world * 30:
            glob = np.concatenate([x_q, x_q[::-1]])[: world * 30]
            dg, ig = O.kneighbors(x_ref, glob, 4, "expanded")
            np.testing.assert_array_equal(r["i_cyc"], ig)
            np.testing.assert_array_equal(r["d_cyc"], dg)


def _ref_worker(rank, world, port, out_dir):
    """Reference-row sharding on CPU: the oracle is each rank's local engine and the merge is the oracle's restatement of
    the device merge (oracle.merge_shards); the distributed plumbing -- shard bounds, global index offsets, the all-gather
    layout (shard, row, slot), the k + 1 candidates of the X=None path -- is the product's RefShardedKNN."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from oracle import oracle as O
    from sknnr_amd.distributed import RefShardedKNN

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x_ref, x_q = _ref_sharded_problem()
        res = {}
        for formula in ("expanded", "direct"):
            for det in (True, False):
                replays = []

                def cand(X, kk, a, b, formula=formula):
                    Xq = x_ref if X is None else X
                    return O.shard_candidates(x_ref[a:b], Xq, kk, formula, index_offset=a)

                def merge(X, k, sv, si, use_det, formula=formula, replays=replays):
                    d, i, n_replay = O.merge_shards(x_ref, X, sv, si, k, formula, deterministic=use_det)
                    replays.append(n_replay)
                    return d, i

                sh = RefShardedKNN(None, local_candidates=cand, merge=merge, n_ref=len(x_ref))
                d, i = sh.kneighbors(x_q, 4, use_deterministic_ordering=det)
                ds, is_ = sh.kneighbors(None, 4, use_deterministic_ordering=det)
                tag = f"{formula}_{int(det)}"
                res.update({f"d_{tag}": d, f"i_{tag}": i, f"ds_{tag}": ds, f"is_{tag}": is_, f"replays_{tag}": np.asarray(replays)})
        # more neighbours than the smallest shard holds: EVERY rank refuses before the collective (a rank that
        # raised alone would leave the others blocked in the all-gather)
        try:
            sh.kneighbors(x_q, len(x_ref) // world + 1)
            res["refused"] = np.asarray(False)
        except ValueError as err:
            res["refused"] = np.asarray("smallest" in str(err))
        np.savez(os.path.join(out_dir, f"ref_rank{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


def _ref_sharded_problem():
    """300 reference rows with exact duplicates placed on BOTH sides of the shard boundaries (rows 150 / 100, 200) and
    queries that are copies of such rows: ties across the k-th slot whose rows live in different shards."""
    from sknnr_amd import synth

    x_ref, _, x_q = synth.make_problem(300, 40, 6, t=2, n_dup_refs=0, n_dup_queries=0)
    for a, b in ((10, 160), (11, 161), (12, 162), (13, 163), (14, 290), (14, 105), (99, 100), (149, 150), (149, 151), (5, 205), (5, 206)):
        x_ref[b] = x_ref[a]
    x_q[:8] = x_ref[[10, 14, 99, 149, 5, 160, 206, 290]]   # zero distances, duplicates split across shards
    x_q[8] = 0.5 * (x_ref[20] + x_ref[170])                  # ordinary rows
    return x_ref, x_q


@pytest.mark.parametrize("world", [2, 3])
def test_reference_sharded_search_matches_the_unsharded_call(tmp_path, world):
    import torch.multiprocessing as mp

    from oracle import oracle as O

    mp.spawn(_ref_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    x_ref, x_q = _ref_sharded_problem()
    for formula in ("expanded", "direct"):
        for det in (True, False):
            d, i = O.kneighbors(x_ref, x_q, 4, formula, deterministic=det)
            ds, is_ = O.kneighbors(x_ref, None, 4, formula, deterministic=det)
            tag = f"{formula}_{int(det)}"
            for rank in range(world):
                r = np.load(os.path.join(str(tmp_path), f"ref_rank{rank}.npz"))
                assert bool(r["refused"])
                np.testing.assert_array_equal(r[f"i_{tag}"], i)
                np.testing.assert_array_equal(r[f"d_{tag}"], d)
                np.testing.assert_array_equal(r[f"is_{tag}"], is_)
                np.testing.assert_array_equal(r[f"ds_{tag}"], ds)
                if formula == "expanded":
                    assert r[f"replays_{tag}"].sum() > 0   # the tied rows really took the replay path
                else:
                    assert r[f"replays_{tag}"].sum() == 0  # (value, index) order is the direct formula's own rule


def test_shard_bounds_partition_the_rows():
    from sknnr_amd.distributed import shard_bounds

    for n in (0, 1, 7, 8, 9, 10_000_000, 50_000_001):
        for w in (1, 2, 4, 8):
            edges = [shard_bounds(n, w, r) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(w - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_needs_an_initialised_group():
    from sknnr_amd.distributed import ShardedKNN

    with pytest.raises(RuntimeError, match="not initialised"):
        ShardedKNN(None)
