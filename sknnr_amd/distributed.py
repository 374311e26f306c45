"""Row-sharded kneighbors / predict over the GPUs of one node.

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on ROCm,
``gloo`` in the CPU tests).  Every rank holds the same fitted estimator (the reference set
is tiny next to 288 GB of HBM: 50k x 64 float64 = 25.6 MB) and answers a *contiguous* block
of the query rows, carrying the block's global row offset so that sknnr's position-dependent
tie-break (``|idx - row|``, /root/reference/src/sknnr/_base.py:171) is identical to the
single-call result.  Each shard's result is final -- the only exchange step is the
all-gather of the per-shard ``(dist, idx)`` blocks (and predictions); the "merge" is
concatenation in rank order.

xGMI is point to point (7 links per GPU): every rank contributes one block and receives
W-1 blocks, one per link, so the gather of chunk ``i`` is issued on a side stream and
overlaps the kernels of chunk ``i+1``; nothing in the data path is a ring of dependent hops.
"""

from __future__ import annotations

from typing import Callable

import numpy as np

__all__ = ["shard_bounds", "cyclic_slot", "all_gather_rows", "ShardedKNN", "RefShardedKNN"]


def shard_bounds(n_rows: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced block of ``rank``: the first ``n % W`` ranks get one extra row."""
    base, rem = divmod(int(n_rows), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def cyclic_slot(world_size: int, rank: int, a: int, b: int) -> int:
    """Chunk-cyclic dealing of a job's rows: every rank holds ``n_local`` rows, cut at the same
    positions ``[a, b)`` into chunks; chunk ``[a, b)`` of all ranks together forms the contiguous
    global rows ``[W*a, W*b)`` with rank ``r``'s block at ``W*a + r*(b-a)`` (returned).  A finished
    chunk can then be all-gathered *in place* into the final arrays."""
    return int(world_size) * int(a) + int(rank) * (int(b) - int(a))


def _max_rows(n_rows: int, world_size: int) -> int:
    return -(-int(n_rows) // int(world_size))


def all_gather_rows(local, n_rows_total: int, group=None, async_op: bool = False):
    """Concatenate the ranks' row blocks (sizes given by :func:`shard_bounds`) on every rank.

    ``local`` is a torch tensor (CUDA under nccl, CPU under gloo) whose rows are this rank's
    block.  Blocks are padded to the common maximum for the collective and trimmed after.
    Returns the gathered tensor, or ``(finish, work)`` when ``async_op`` is set, where
    ``finish()`` waits and returns it.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    pad_rows = _max_rows(n_rows_total, world)
    tail = tuple(local.shape[1:])
    send = local
    if local.shape[0] != pad_rows:
        send = torch.zeros((pad_rows, *tail), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    recv = torch.empty((world * pad_rows, *tail), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()
        if n_rows_total == world * pad_rows:
            return recv
        parts = []
        for r in range(world):
            a, b = shard_bounds(n_rows_total, world, r)
            parts.append(recv[r * pad_rows : r * pad_rows + (b - a)])
        return torch.cat(parts, dim=0)

    return (finish, work) if async_op else finish()


class ShardedKNN:
    """Query-row sharding of a fitted estimator across the ranks of ``group``.

    Parameters
    ----------
    estimator : a fitted ``sknnr_amd`` estimator (every rank fits/loads the same one), or None
        together with explicit ``local_kneighbors`` / ``local_predict`` callables.
    local_kneighbors : ``f(X_block, row_offset, n_neighbors, **kw) -> (dist, idx)`` override
        (the CPU tests plug the oracle in here; the product path uses the estimator's engine).
    local_predict : ``f(X_block, row_offset) -> pred`` override.
    """

    def __init__(self, estimator=None, group=None, local_kneighbors: Callable | None = None,
                 local_predict: Callable | None = None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
        self.estimator = estimator
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self._local_kneighbors = local_kneighbors or self._engine_kneighbors
        self._local_predict = local_predict or self._engine_predict
        self._local_out = local_kneighbors is None  # the engine can write into caller tensors

    # ---- local engines ------------------------------------------------------------------
    def _reg(self):
        est = self.estimator
        return getattr(est, "regressor_", est), hasattr(est, "regressor_")

    def _engine_kneighbors(self, X_block, row_offset, n_neighbors, use_deterministic_ordering=True,
                           n_self_rows=None, out=None):
        reg, transformed = self._reg()
        k = reg._resolve_k(n_neighbors)
        return reg._kneighbors_engine(X_block, k, apply_affine=transformed and X_block is not None,
                                      use_deterministic_ordering=use_deterministic_ordering,
                                      row_offset=row_offset, n_self_rows=n_self_rows, out=out)

    def _engine_predict(self, X_block, row_offset, n_self_rows=None):
        reg, transformed = self._reg()
        return reg._predict_engine(X_block, apply_affine=transformed and X_block is not None,
                                   row_offset=row_offset, n_self_rows=n_self_rows)

    # ---- helpers --------------------------------------------------------------------------
    def local_bounds(self, n_rows: int) -> tuple[int, int]:
        return shard_bounds(n_rows, self.world_size, self.rank)

    def _as_comm_tensor(self, a):
        import torch
        import torch.distributed as dist

        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a))
        if dist.get_backend(self.group) == "nccl" and not t.is_cuda:
            t = t.cuda()
        return t

    # ---- public ---------------------------------------------------------------------------
    def kneighbors(self, X=None, n_neighbors=None, *, n_rows_total=None, X_is_local_block=False,
                   use_deterministic_ordering=True, gather=True, **kw):
        """Neighbours of all query rows, computed one block per rank.

        ``X`` is either the whole query matrix (every rank slices its block) or, with
        ``X_is_local_block=True``, already this rank's block of a call of ``n_rows_total`` rows.
        ``X=None`` shards the fit-time self query (the reference's ``X=None`` path) over the
        reference rows.  With ``gather=False`` only the local block is returned.
        """
        if X is None:
            n_rows_total = self._reg()[0].n_samples_fit_ if n_rows_total is None else n_rows_total
            a, b = self.local_bounds(n_rows_total)
            dist_l, idx_l = self._local_kneighbors(None, a, n_neighbors, n_self_rows=b - a,
                                                   use_deterministic_ordering=use_deterministic_ordering, **kw)
        else:
            if X_is_local_block:
                if n_rows_total is None:
                    raise ValueError("n_rows_total is required with X_is_local_block=True")
                a, b = self.local_bounds(n_rows_total)
                block = X
                if block.shape[0] != b - a:
                    raise ValueError(f"rank {self.rank} expects {b - a} rows, got {block.shape[0]}")
            else:
                n_rows_total = X.shape[0]
                a, b = self.local_bounds(n_rows_total)
                block = X[a:b]
            dist_l, idx_l = self._local_kneighbors(block, a, n_neighbors,
                                                   use_deterministic_ordering=use_deterministic_ordering, **kw)
        if not gather or self.world_size == 1:
            return dist_l, idx_l
        as_numpy = isinstance(idx_l, np.ndarray)
        d_all = all_gather_rows(self._as_comm_tensor(dist_l), n_rows_total, self.group)
        i_all = all_gather_rows(self._as_comm_tensor(idx_l), n_rows_total, self.group)
        if as_numpy:
            return d_all.cpu().numpy(), i_all.cpu().numpy()
        return d_all, i_all

    def predict(self, X=None, *, n_rows_total=None, X_is_local_block=False, gather=True):
        """Predictions of all query rows, one block per rank, all-gathered."""
        if X is None:
            n_rows_total = self._reg()[0].n_samples_fit_ if n_rows_total is None else n_rows_total
            a, b = self.local_bounds(n_rows_total)
            pred_l = self._local_predict(None, a, n_self_rows=b - a)
        else:
            if X_is_local_block:
                if n_rows_total is None:
                    raise ValueError("n_rows_total is required with X_is_local_block=True")
                a, b = self.local_bounds(n_rows_total)
                block = X
            else:
                n_rows_total = X.shape[0]
                a, b = self.local_bounds(n_rows_total)
                block = X[a:b]
            pred_l = self._local_predict(block, a)
        if not gather or self.world_size == 1:
            return pred_l
        as_numpy = isinstance(pred_l, np.ndarray)
        p_all = all_gather_rows(self._as_comm_tensor(pred_l), n_rows_total, self.group)
        return p_all.cpu().numpy() if as_numpy else p_all

    def kneighbors_pipelined(self, X_block, n_rows_total, n_neighbors=None, *, chunk_rows=1 << 21,
                             use_deterministic_ordering=True):
        """Device-resident block in, gathered ``(dist, idx)`` of the whole call out, with the
        all-gather of chunk ``i`` overlapping the kernels of chunk ``i+1`` (nccl/RCCL only).

        Chunks are cut at the same row positions on every rank (blocks are padded to the common
        maximum), so every collective has equal sizes everywhere.
        """
        import torch

        a, b = self.local_bounds(n_rows_total)
        n_local = b - a
        pad_rows = _max_rows(n_rows_total, self.world_size)
        reg = self._reg()[0]
        k = reg._resolve_k(n_neighbors)
        dev = X_block.device
        comm = torch.cuda.Stream(device=dev)
        compute = torch.cuda.current_stream(dev)
        pending = []
        for c0 in range(0, pad_rows, chunk_rows):
            c1 = min(pad_rows, c0 + chunk_rows)
            lo, hi = min(c0, n_local), min(c1, n_local)
            d_c = torch.zeros((c1 - c0, k), dtype=torch.float64, device=dev)
            i_c = torch.zeros((c1 - c0, k), dtype=torch.int64, device=dev)
            if hi > lo:
                d_l, i_l = self._local_kneighbors(X_block[lo:hi], a + lo, k,
                                                  use_deterministic_ordering=use_deterministic_ordering)
                d_c[: hi - lo] = d_l
                i_c[: hi - lo] = i_l
            ready = torch.cuda.Event()
            ready.record(compute)
            with torch.cuda.stream(comm):
                comm.wait_event(ready)
                d_all = torch.empty((self.world_size * (c1 - c0), k), dtype=torch.float64, device=dev)
                i_all = torch.empty((self.world_size * (c1 - c0), k), dtype=torch.int64, device=dev)
                import torch.distributed as dist

                w1 = dist.all_gather_into_tensor(d_all, d_c, group=self.group, async_op=True)
                w2 = dist.all_gather_into_tensor(i_all, i_c, group=self.group, async_op=True)
            pending.append((c0, c1, d_c, i_c, d_all, i_all, w1, w2))
        out_d = torch.empty((n_rows_total, k), dtype=torch.float64, device=dev)
        out_i = torch.empty((n_rows_total, k), dtype=torch.int64, device=dev)
        for c0, c1, _d_c, _i_c, d_all, i_all, w1, w2 in pending:
            w1.wait()
            w2.wait()
            rows = c1 - c0
            for r in range(self.world_size):
                ra, rb = shard_bounds(n_rows_total, self.world_size, r)
                lo, hi = min(c0, rb - ra), min(c1, rb - ra)
                if hi > lo:
                    out_d[ra + lo : ra + hi] = d_all[r * rows : r * rows + (hi - lo)]
                    out_i[ra + lo : ra + hi] = i_all[r * rows : r * rows + (hi - lo)]
        compute.wait_stream(comm)
        return out_d, out_i

    def kneighbors_cyclic(self, X_block, n_neighbors=None, *, chunk_rows=2_500_000,
                          use_deterministic_ordering=True):
        """Rows dealt chunk-cyclically (:func:`cyclic_slot`; every rank passes the same number of
        rows): each rank writes a finished chunk straight into its slot of the final
        ``(W*n_local, k)`` arrays and the all-gather of that chunk is in place -- no staging
        copies -- and, under nccl/RCCL, issued on a side stream so that it travels under the next
        chunk's kernels.  Returns the complete ``(dist, idx)`` in global row order on every rank."""
        import torch
        import torch.distributed as dist

        reg = self._reg()[0] if self.estimator is not None else None
        k = reg._resolve_k(n_neighbors) if reg is not None else int(n_neighbors)
        W, r = self.world_size, self.rank
        n_local = X_block.shape[0]
        on_gpu = isinstance(X_block, torch.Tensor) and X_block.is_cuda
        dev = X_block.device if on_gpu else torch.device("cpu")
        d_all = torch.empty((W * n_local, k), dtype=torch.float64, device=dev)
        i_all = torch.empty((W * n_local, k), dtype=torch.int64, device=dev)
        comm = torch.cuda.Stream(device=dev) if on_gpu else None
        works = []
        for a in range(0, n_local, chunk_rows):
            b = min(n_local, a + chunk_rows)
            lo = cyclic_slot(W, r, a, b)
            d_own, i_own = d_all[lo: lo + (b - a)], i_all[lo: lo + (b - a)]
            d_l, i_l = self._local_kneighbors(X_block[a:b], lo, k,
                                              use_deterministic_ordering=use_deterministic_ordering,
                                              **({"out": (d_own, i_own)} if self._local_out else {}))
            if not self._local_out:
                d_own.copy_(torch.as_tensor(d_l))
                i_own.copy_(torch.as_tensor(i_l))
            if W == 1:
                continue
            if on_gpu:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(dev))
                with torch.cuda.stream(comm):
                    comm.wait_event(done)
                    works.append(dist.all_gather_into_tensor(d_all[W * a: W * b], d_own, group=self.group, async_op=True))
                    works.append(dist.all_gather_into_tensor(i_all[W * a: W * b], i_own, group=self.group, async_op=True))
            else:
                works.append(dist.all_gather_into_tensor(d_all[W * a: W * b], d_own, group=self.group, async_op=True))
                works.append(dist.all_gather_into_tensor(i_all[W * a: W * b], i_own, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if on_gpu:
            torch.cuda.current_stream(dev).wait_stream(comm)
        return d_all, i_all


class RefShardedKNN:
    """REFERENCE-row sharding ("alternative"): rank ``r`` sweeps rows ``shard_bounds(n_ref, W, r)``
    of the reference set for ALL query rows, the ranks all-gather their per-shard candidates ``(value, index)`` --
    ``k`` (+1 for X=None) pairs per query and shard -- and every rank merges them into the call's answer.  The mode for
    few queries against a large reference set (the query-row sharding of :class:`ShardedKNN` needs at least a
    workgroup of rows per GPU to keep them busy).  The reference's analogue is scikit-learn's parallel-on-Y strategy
    (per-thread heaps over chunks of Y, then ``_parallel_on_Y_synchronize``:
    scikit-learn's ArgKmin, _argkmin.pyx.tp:200-261).

    Exactness: the merged list (smallest (value, index) first) is the answer whenever it is unique; rows with an exact
    tie across the last slot are re-scanned over all reference rows by the full engine every rank holds (the small
    float64 copy of the rows; only the SWEEP is split), exactly as the unsharded call treats tied rows -- a sharded
    call returns what the unsharded call returns.

    Parameters
    ----------
    estimator : a fitted ``sknnr_amd`` estimator (every rank the same one), or None with the three callables
    local_candidates : ``f(X, kk, a, b) -> (val (nq, kk), idx (nq, kk))`` candidates of reference rows [a, b)
    merge : ``f(X, k, shard_val, shard_idx, use_deterministic_ordering) -> (dist, idx)``
    n_ref : number of reference rows (with the callables)
    """

    def __init__(self, estimator=None, group=None, local_candidates: Callable | None = None,
                 merge: Callable | None = None, n_ref: int | None = None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
        self.estimator = estimator
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self._shard_engine = None
        if estimator is not None:
            reg = getattr(estimator, "regressor_", estimator)
            self.n_ref = reg.n_samples_fit_
        else:
            if local_candidates is None or merge is None or n_ref is None:
                raise ValueError("without an estimator, local_candidates, merge and n_ref are required")
            self.n_ref = int(n_ref)
        self._local_candidates = local_candidates or self._engine_candidates
        self._merge = merge or self._engine_merge
        self.bounds = shard_bounds(self.n_ref, self.world_size, self.rank)

    # ---- the HIP engine as local worker ---------------------------------------------------------
    def _reg(self):
        est = self.estimator
        return getattr(est, "regressor_", est), hasattr(est, "regressor_")

    def _shard(self):
        """Engine over this rank's rows of the (transformed) reference set, with the estimator's affine map."""
        if self._shard_engine is None:
            from ._engine import KNNEngine

            reg, _ = self._reg()
            a, b = self.bounds
            eng = KNNEngine(reg._fit_X[a:b], None, device=reg._device)
            if getattr(reg, "effective_metric_", "euclidean") == "hamming":
                eng.set_hamming_weights(reg._hamming_w)
            if reg._affine is not None:
                eng.set_affine(*reg._affine)
            self._shard_engine = eng
        return self._shard_engine

    def _engine_candidates(self, X, kk, a, b):
        reg, transformed = self._reg()
        if b - a < kk:
            raise ValueError(f"rank {self.rank} holds {b - a} reference rows, fewer than the {kk} neighbours asked for")
        if X is None:  # the X=None path: the queries are the (transformed) reference rows themselves
            return self._shard().shard_candidates(reg._fit_X, kk, formula=reg._formula(), index_offset=a)
        return self._shard().shard_candidates(X, kk, formula=reg._formula(), apply_affine=self._device_affine(), index_offset=a,
                                              check_finite=True)

    def _engine_merge(self, X, k, shard_val, shard_idx, use_deterministic_ordering):
        reg, transformed = self._reg()
        return reg.engine_.merge_shards(X, k, shard_val, shard_idx, exclude_self=X is None,
                                        deterministic=use_deterministic_ordering,
                                        decimals=reg.DISTANCE_PRECISION_DECIMALS, formula=reg._formula(),
                                        apply_affine=self._device_affine() and X is not None)

    def _device_affine(self):
        """Is the estimator's feature map applied on the device (affine spaces) or were the rows mapped on the host
        (tree-node spaces)?"""
        _, transformed = self._reg()
        return transformed and getattr(self.estimator, "_device_affine", True)

    # ---- public -----------------------------------------------------------------------------------
    def kneighbors(self, X=None, n_neighbors=None, *, use_deterministic_ordering=True):
        """Neighbours of every row of ``X`` (every rank passes the same rows; ``None``: of every reference row, itself
        excluded), the reference rows swept shard by shard."""
        import torch
        import torch.distributed as dist

        if self.estimator is not None:
            reg, transformed = self._reg()
            k = reg._resolve_k(n_neighbors)
            if X is not None:
                X = (self.estimator._validate_raw_query(X) if transformed else reg._validate_query(X))
        else:
            k = int(n_neighbors)
        kk = k + (1 if X is None else 0)
        a, b = self.bounds
        # The same test with the same outcome on EVERY rank, before any collective: shard sizes differ by at most one row, so
        # the smallest shard decides -- a rank that raised alone would leave the others blocked in the all-gather.
        smallest = self.n_ref // self.world_size
        if smallest < kk:
            raise ValueError(f"the smallest of the {self.world_size} reference shards holds {smallest} rows, fewer than the {kk} "
                             "neighbours asked for of every shard")
        val, idx = self._local_candidates(X, kk, a, b)
        as_numpy = isinstance(idx, np.ndarray)
        tv = val if isinstance(val, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(val))
        ti = idx if isinstance(idx, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(idx))
        if dist.get_backend(self.group) == "nccl" and not tv.is_cuda:
            tv, ti = tv.cuda(), ti.cuda()
        tv, ti = tv.contiguous(), ti.contiguous()
        nq = tv.shape[0]
        all_v = torch.empty((self.world_size * nq, kk), dtype=tv.dtype, device=tv.device)
        all_i = torch.empty((self.world_size * nq, kk), dtype=ti.dtype, device=ti.device)
        # one exchange step: every rank contributes (nq, kk) pairs and receives W - 1 such blocks, one per xGMI link
        dist.all_gather_into_tensor(all_v, tv, group=self.group)
        dist.all_gather_into_tensor(all_i, ti, group=self.group)
        all_v, all_i = all_v.view(self.world_size, nq, kk), all_i.view(self.world_size, nq, kk)
        if as_numpy:
            all_v, all_i = all_v.cpu().numpy(), all_i.cpu().numpy()
        return self._merge(X, k, all_v, all_i, use_deterministic_ordering)
