"""Shared test configuration.

``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI load/export checks, gloo
sharding -- runs anywhere.  ``-m gpu``: parity of the HIP path against the oracle and the
golden vectors, through the C ABI -- needs an MI355X.
"""

from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def moscow():
    """Moscow Mountain / St. Joes with the reference tests' fixed 80/20 unshuffled split
    (/root/reference/tests/conftest.py:45-59): 132 train / 33 test rows."""
    from sknnr_amd.datasets import load_moscow_stjoes

    ds = load_moscow_stjoes()
    n_train = 132
    return {
        "X_train": ds.data[:n_train], "X_test": ds.data[n_train:],
        "y_train": ds.target[:n_train], "y_test": ds.target[n_train:],
        "index": ds.index, "feature_names": ds.feature_names, "target_names": ds.target_names,
    }


@pytest.fixture(scope="session")
def moscow_frames():
    from sknnr_amd.datasets import load_moscow_stjoes

    X, y = load_moscow_stjoes(return_X_y=True, as_frame=True)
    return {"X_train": X.iloc[:132], "X_test": X.iloc[132:], "y_train": y.iloc[:132], "y_test": y.iloc[132:],
            "X_all": X, "y_all": y}


def yaimpute_weights(d):
    """The callable the reference's regression tests use (tests/test_regressions.py:31-39)."""
    return 1.0 / (1.0 + d)


def assert_neighbors_match(idx, dist, ref_idx, ref_dist, fit_X=None, rtol=1e-5, atol=1e-8):
    """Indices must be identical, except where the reference itself is arbitrary: two
    *bitwise identical* reference rows tied for a slot (the pick then depends on OpenBLAS
    kernel position effects / kd-tree visiting order).  Distances within the reference's own
    regression tolerance (pytest-regressions defaults rtol 1e-5, atol 1e-8)."""
    idx, ref_idx = np.asarray(idx), np.asarray(ref_idx)
    np.testing.assert_allclose(np.asarray(dist), np.asarray(ref_dist), rtol=rtol, atol=atol)
    bad = idx != ref_idx
    if not bad.any():
        return 0
    assert fit_X is not None, f"{int(bad.sum())} neighbour indices differ"
    rows, cols = np.nonzero(bad)
    for r, c in zip(rows, cols):
        a, b = int(idx[r, c]), int(ref_idx[r, c])
        same_row = np.array_equal(fit_X[a], fit_X[b])
        # the two answers may also hold the same tied set in a different order
        same_set = sorted(idx[r].tolist()) == sorted(ref_idx[r].tolist())
        if not (same_row or same_set):
            raise AssertionError(f"row {r}: neighbour {c} is {a}, reference says {b} (rows differ)")
        if same_set and not same_row:
            raise AssertionError(f"row {r}: same neighbours in a different order: {idx[r]} vs {ref_idx[r]}")
    return int(bad.sum())


def assert_hamming_neighbors_match(idx, dist, ref_idx, ref_dist, fit_ids, q_ids, w, row_offset_self=None,
                                   rtol=1e-5, atol=1e-8):
    """Weighted-Hamming results against the reference's: distances must agree (as sorted rows: bitwise
    equal distances may be listed in either order of their rows); indices must agree except inside the
    class of reference rows tied EXACTLY at the k-th distance -- numpy's argpartition picks among those
    by the internals of its introselect, which is not restated ("Hamming ties").  Every
    returned index must really be at its reported distance, and nothing strictly closer may be missing.
    Which tied row argpartition keeps is a property of the machine, not of the algorithm: numpy dispatches it to
    x86-simd-sort's vectorised quick-select on AVX2 / AVX-512 hosts and to the scalar introselect elsewhere, and
    the two keep different rows on these very fixtures (scripts/hamming_tie_dispatch.py,
    profiles/r03_hamming_tie_dispatch.txt).
    Returns the (sorted) row numbers whose index sets differ -- all of them boundary-tie rows; callers compare
    everything downstream (predictions, scores) on the complement."""
    from scipy.spatial.distance import cdist

    idx, ref_idx = np.asarray(idx), np.asarray(ref_idx)
    dist, ref_dist = np.asarray(dist), np.asarray(ref_dist)
    np.testing.assert_allclose(np.sort(dist, axis=1), np.sort(ref_dist, axis=1), rtol=rtol, atol=atol)
    full = cdist(np.asarray(q_ids, dtype=np.float64), np.asarray(fit_ids, dtype=np.float64), "hamming", w=w)
    differing = []
    for r in range(idx.shape[0]):
        np.testing.assert_allclose(full[r, idx[r]], dist[r], rtol=rtol, atol=atol)  # honest distances
        if sorted(idx[r].tolist()) == sorted(ref_idx[r].tolist()):
            continue
        differing.append(r)
        kth = ref_dist[r].max()
        only_mine = set(idx[r].tolist()) - set(ref_idx[r].tolist())
        only_ref = set(ref_idx[r].tolist()) - set(idx[r].tolist())
        for j in only_mine | only_ref:
            assert full[r, j] == pytest.approx(kth, rel=rtol, abs=atol), (
                f"row {r}: index {j} at distance {full[r, j]} is not tied with the k-th distance {kth}")
        row = full[r].copy()
        if row_offset_self is not None:
            row[row_offset_self + r] = np.inf
        assert (np.sort(row)[: idx.shape[1]].max() <= kth + atol + rtol * abs(kth)), f"row {r}: a closer row is missing"
        # ... and the row really is a boundary-tie row: the k-th and (k+1)-th smallest distances are equal
        srt = np.sort(row)
        assert srt[idx.shape[1] - 1] == pytest.approx(srt[idx.shape[1]], rel=rtol, abs=atol), f"row {r}: sets differ without a tie"
    return np.asarray(differing, dtype=np.int64)


def rows_without(n_rows, tie_rows):
    """Boolean mask of the rows that are NOT in ``tie_rows``."""
    keep = np.ones(n_rows, dtype=bool)
    keep[np.asarray(tie_rows, dtype=np.int64)] = False
    return keep


def mixed_forest_y_fit(y_train):
    """The ``y_fit`` of the reference's mixed regression / classification forest test
    (/root/reference/tests/test_regressions.py:157-176): Total_BA (numeric -> regression forest) and the
    species of maximum basal area reclassed to ABGR_BA / TSHE_BA / OTHER (strings -> classification forest)."""
    cols = [c for c in y_train.columns if c.endswith("_BA") and c != "Total_BA"]
    max_species = y_train[cols].idxmax(axis=1)
    max_species = max_species.where(max_species.isin(["ABGR_BA", "TSHE_BA"]), other="OTHER")
    return y_train[["Total_BA"]].assign(MAX_SPECIES=max_species)
