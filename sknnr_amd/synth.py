"""Seeded synthetic inputs for the kneighbors/predict hot path.

The generators follow SURVEY.md §8(d): correlated Gaussian features
``X = Z @ (I + 0.3 G)`` and multi-output targets that are (log-)linear in the
standardised features, so that CCA reaches ``D_t = D_in`` components and CCorA
finds ``>= 8`` significant variates.  They are shared by ``bench.py``, the
golden-vector generator (``tests/golden/make_golden.py``) and the parity tests,
so every consumer sees bit-identical inputs for a given seed.
"""

from __future__ import annotations

import numpy as np

__all__ = ["mixing_matrix", "make_features", "make_targets", "make_problem"]


def mixing_matrix(d: int, seed: int = 0) -> np.ndarray:
    """(d, d) matrix ``I + 0.3 G`` with ``G ~ N(0, 1)``, seeded."""
    rng = np.random.default_rng([seed, 7919, d])
    return np.eye(d) + 0.3 * rng.standard_normal((d, d))


def make_features(n: int, d: int, seed: int, mix_seed: int = 0) -> np.ndarray:
    """``n`` rows of correlated features, float64, C-contiguous."""
    rng = np.random.default_rng([seed, n, d])
    z = rng.standard_normal((n, d))
    return np.ascontiguousarray(z @ mixing_matrix(d, mix_seed))


def make_targets(
    x: np.ndarray, t: int = 40, seed: int = 2, kind: str = "linear"
) -> np.ndarray:
    """Targets for ``x``: ``kind='linear'`` (CCorA / Mahalanobis / Euclidean) or
    ``kind='positive'`` (strictly positive rows, for CCA)."""
    n, d = x.shape
    rng = np.random.default_rng([seed, n, d, t])
    w = rng.standard_normal((d, t))
    z = (x - x.mean(axis=0)) / x.std(axis=0)
    lin = z @ w / np.sqrt(d)
    noise = rng.standard_normal((n, t))
    if kind == "linear":
        return lin + 0.5 * noise
    if kind == "positive":
        return np.exp(0.5 * lin + 0.1 * noise)
    raise ValueError(f"unknown target kind {kind!r}")


def make_problem(
    n_ref: int,
    n_query: int,
    d: int,
    t: int = 40,
    kind: str = "linear",
    n_dup_refs: int = 0,
    n_dup_queries: int = 0,
):
    """Reference rows, targets and query rows.

    ``n_dup_refs`` copies the first rows of the reference set over its last rows
    (exact duplicate references -> exact distance ties); ``n_dup_queries`` makes the
    first queries exact copies of reference rows (zero distances).
    """
    x_ref = make_features(n_ref, d, seed=0)
    x_q = make_features(n_query, d, seed=1)
    if n_dup_refs:
        x_ref[n_ref - n_dup_refs :] = x_ref[:n_dup_refs]
    if n_dup_queries:
        step = max(1, n_ref // n_dup_queries)
        x_q[:n_dup_queries] = x_ref[::step][:n_dup_queries]
    y = make_targets(x_ref, t=t, kind=kind)
    return x_ref, y, x_q


def make_forest_ids(n_ref: int, nq: int, n_trees: int, d: int = 6, cuts: int = 7, seed: int = 0, query_noise: float = 0.15):
    """Node ids shaped like a forest's ``apply`` output, without growing one: points in ``d`` dimensions (queries are
    jittered copies of reference points), and per tree an axis-aligned partition of a random 3-D subspace into
    ``cuts``^3 cells at jittered quantiles -- rows that are close share the cell in most trees, rows that are not share it
    in about 1 / cuts^3 of them.  Returns float64 ``(ref_ids, query_ids)`` of shape ``(n, n_trees)``, ids < cuts^3."""
    rng = np.random.default_rng(seed)
    x_ref = rng.standard_normal((n_ref, d))
    src = rng.integers(0, n_ref, nq)
    x_q = x_ref[src] + query_noise * rng.standard_normal((nq, d))
    ref_ids = np.empty((n_ref, n_trees), dtype=np.float64)
    q_ids = np.empty((nq, n_trees), dtype=np.float64)
    base = np.linspace(0.0, 1.0, cuts + 1)[1:-1]
    for t in range(n_trees):
        feats = rng.choice(d, size=3, replace=False)
        rid = np.zeros(n_ref, dtype=np.int64)
        qid = np.zeros(nq, dtype=np.int64)
        for f in feats:
            edges = np.quantile(x_ref[:, f], np.clip(base + rng.uniform(-0.4, 0.4, cuts - 1) / cuts, 0.01, 0.99))
            edges.sort()
            rid = rid * cuts + np.searchsorted(edges, x_ref[:, f])
            qid = qid * cuts + np.searchsorted(edges, x_q[:, f])
        ref_ids[:, t] = rid
        q_ids[:, t] = qid
    return ref_ids, q_ids
