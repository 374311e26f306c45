"""Process-wide settings of the MI355X backend that have no counterpart among the reference's constructor
parameters (so they cannot live there without changing the estimators' public signature)."""

from __future__ import annotations

import contextlib
import os

__all__ = ["get_hamming_tie_policy", "set_hamming_tie_policy", "hamming_tie_policy"]

_HAMMING_TIE_POLICIES = ("lowest_index", "numpy")
_hamming_tie_policy = os.environ.get("SKNNR_HAMMING_TIES", "lowest_index")
if _hamming_tie_policy not in _HAMMING_TIE_POLICIES:
    raise ValueError(f"SKNNR_HAMMING_TIES must be one of {_HAMMING_TIE_POLICIES}, got {_hamming_tie_policy!r}")


def get_hamming_tie_policy() -> str:
    return _hamming_tie_policy


def set_hamming_tie_policy(policy: str) -> None:
    """Which reference rows RFNN / GBNN keep when several are tied EXACTLY at the k-th weighted-Hamming distance.

    ``"lowest_index"`` (default)
        the device's rule: among tied rows the lowest reference index first.  Deterministic on every machine.
    ``"numpy"``
        the reference's rule: whatever ``np.argpartition`` keeps (REF src/sknnr/_weighted_trees.py:53-59 ->
        SKL/neighbors/_base.py:733-760, ``_kneighbors_reduce_func``).  Rows with a tie are detected on the device
        results; their full float64 distance rows come back from the device (``sknnr_hamming_distances``) and the
        selection is replayed with this host's numpy, so the result equals the reference run on the same host --
        including the reference's committed regression files on x86 hosts (both the AVX-512 and the AVX2 dispatch of
        numpy reproduce them; profiles/r03_hamming_tie_dispatch.txt).  Costs a second search plus one distance row
        per tied query.
    """
    global _hamming_tie_policy
    if policy not in _HAMMING_TIE_POLICIES:
        raise ValueError(f"hamming tie policy must be one of {_HAMMING_TIE_POLICIES}, got {policy!r}")
    _hamming_tie_policy = policy


@contextlib.contextmanager
def hamming_tie_policy(policy: str):
    """``with sknnr_amd.hamming_tie_policy("numpy"): est.fit(...); est.kneighbors(...)``"""
    before = get_hamming_tie_policy()
    set_hamming_tie_policy(policy)
    try:
        yield
    finally:
        set_hamming_tie_policy(before)
