#!/usr/bin/env python
"""Is the reference's pick among rows tied at the k-th weighted-Hamming distance a property of the
algorithm, or of the machine it runs on?  

The reference forces ``algorithm="brute", metric="hamming"`` (REF src/sknnr/_weighted_trees.py:53-59), which
lands in scikit-learn's chunked pairwise path whose reduce step is ``np.argpartition(dist, k - 1, axis=1)``
(SKL/neighbors/_base.py:733-760).  numpy dispatches argpartition at run time: x86-simd-sort's vectorised
quick-select when the CPU has AVX-512 (SKX) or AVX2, the scalar introselect otherwise
(numpy/_core/src/npysort/selection.cpp, ``aquickselect_dispatch``).  This script runs that exact sklearn call
on the node-id matrices / weights of the committed Moscow fixtures in child processes that differ only in
``NPY_DISABLE_CPU_FEATURES`` and compares the neighbour SETS row by row with the reference maintainers'
committed ``.npz`` (tests/golden/ref_regressions).

CPU only; prints a table.  Recorded in profiles/r03_hamming_tie_dispatch.txt.
"""

from __future__ import annotations

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

AVX512 = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"
VARIANTS = [
    ("all features (AVX-512 argselect)", ""),
    ("AVX-512 off (AVX2 argselect)", AVX512),
    ("AVX-512 + AVX2 off (scalar introselect)", "AVX2 FMA3 " + AVX512),
]
CASES = [  # (fixture, the reference maintainers' file for kneighbors(X_test) with array indices)
    ("moscow_rfnn", "test_kneighbors_target_full_randomForest_k5_index_.npz"),
    ("moscow_gbnn", "test_kneighbors_target_full_gbnn_k5_index_.npz"),
    ("moscow_gbnn_uniform", None),
    ("moscow_rfnn_weighted", None),
]

CHILD = r"""
import json, sys
import numpy as np
from sklearn.neighbors import KNeighborsRegressor
g = np.load(sys.argv[1])
tr, te, w = g["ids_train"], g["ids_test"], g["hamming_weights"]
reg = KNeighborsRegressor(n_neighbors=5, algorithm="brute", metric="hamming", metric_params={"w": w})
reg.fit(tr, np.zeros(len(tr)))
d, i = reg.kneighbors(te)
print(json.dumps({"nn": np.sort(i, axis=1).tolist(), "dist": d.tolist()}))
"""


def run_child(fixture: str, disabled: str):
    env = dict(os.environ)
    if disabled:
        env["NPY_DISABLE_CPU_FEATURES"] = disabled
    else:
        env.pop("NPY_DISABLE_CPU_FEATURES", None)
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, "tests", "golden", fixture + ".npz")],
                         env=env, capture_output=True, text=True, check=True)
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    import numpy as np

    print(f"numpy {np.__version__}; host CPU features decide np.argpartition's code path\n")
    for fixture, ref_file in CASES:
        res = [run_child(fixture, dis) for _, dis in VARIANTS]
        base = np.asarray(res[0]["nn"])
        dist = np.asarray(res[0]["dist"])
        g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
        tr, te, w = g["ids_train"], g["ids_test"], g["hamming_weights"]
        # rows with an exact tie across the k-th slot (from the full distance rows, scipy's arithmetic)
        from scipy.spatial.distance import cdist

        full = cdist(te.astype(float), tr.astype(float), "hamming", w=w)
        srt = np.sort(full, axis=1)
        tie_rows = np.nonzero(srt[:, 4] == srt[:, 5])[0]
        print(f"{fixture}: {len(te)} query rows x {len(tr)} reference rows x {tr.shape[1]} trees; "
              f"rows with an exact tie across the 5th slot: {tie_rows.tolist()}")
        committed = None
        if ref_file:
            committed = np.sort(np.load(os.path.join(ROOT, "tests", "golden", "ref_regressions", ref_file))["nn"], axis=1)
        for (label, _), r in zip(VARIANTS, res):
            nn = np.asarray(r["nn"])
            differ = np.nonzero((nn != base).any(axis=1))[0]
            line = f"  {label:42s} rows whose neighbour SET differs from the first variant: {differ.tolist()}"
            if committed is not None:
                dc = np.nonzero((nn != committed).any(axis=1))[0]
                line += f"; from the reference's committed file: {dc.tolist()}"
            assert np.array_equal(np.sort(np.asarray(r["dist"]), axis=1), np.sort(dist, axis=1)), "distances must not depend on the dispatch"
            assert set(differ.tolist()) <= set(tie_rows.tolist()), "sets may differ on tie rows only"
            print(line)
        print()


if __name__ == "__main__":
    main()
