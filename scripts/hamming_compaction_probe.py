"""Development aid: fallbacks of the integer weighted-Hamming pre-filter (candidate list overflow) by neighbours and rows."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N, synth
for n_ref in (20_000, 200_000):
    ids_ref, ids_q = synth.make_forest_ids(n_ref, 512, 60, seed=3)
    for wname, w in (("random", np.random.default_rng(9).random(60) + 0.05), ("equal", np.full(60, 1 / 60))):
        ix = N.Index(ids_ref)
        ix.set_hamming_weights(w)
        for kk in (5, 16, 32):
            ix.reset_stats()
            d, i = ix.kneighbors_host(ids_q, ix.make_opts(kk, formula=N.FORMULA_HAMMING))
            st = ix.stats()
            full = ix.hamming_distances_host(ids_q, np.arange(8))
            srt = np.sort(full, axis=1)
            ties = [(int((row == row[kk - 1]).sum()), int((row <= row[kk - 1]).sum())) for row in srt]
            print(f"n_ref {n_ref} weights {wname} kk {kk}: fallbacks {st['exact_fallbacks']} of 512; (rows tied at the kk-th distance, rows <= it) of 8 queries: {ties}", flush=True)
        ix.close()
