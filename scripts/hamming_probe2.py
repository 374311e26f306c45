#!/usr/bin/env python
"""Development aid (round 3): the weighted-Hamming search with the integer pre-filter (hamming.hip.h) against the
float64 scan (SKNNR_HAMMING_INT=0, read when the weights are installed): device-resident inputs, kernel time from the
handle's events, a slice checked against the oracle.  usage: hamming_probe2.py n_ref nq n_trees k [levels] [uniform]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from sknnr_amd import _native as N  # noqa: E402

n_ref, nq, t, k = (int(a) for a in sys.argv[1:5])
levels = int(sys.argv[5]) if len(sys.argv) > 5 else 300
uniform = len(sys.argv) > 6 and sys.argv[6] == "uniform"
rng = np.random.default_rng(0)
ref = rng.integers(0, levels, (n_ref, t)).astype(np.float64)
q = rng.integers(0, levels, (nq, t)).astype(np.float64)
# trees agree far more often than random ids do: make each query a perturbed copy of a reference row
src = rng.integers(0, n_ref, nq)
keep = rng.random((nq, t)) < 0.6
q = np.where(keep, ref[src], q)
w = np.full(t, 1.0 / t) if uniform else (rng.random(t) + 0.01)
qd = torch.as_tensor(q, device="cuda")
dd = torch.empty((nq, k), dtype=torch.float64, device="cuda")
di = torch.empty((nq, k), dtype=torch.int64, device="cuda")
n_chk = min(nq, 1500)
od, oi = O.kneighbors_hamming(ref, q[:n_chk], w, k)
for mode in ("0", "1"):
    os.environ["SKNNR_HAMMING_INT"] = mode
    ix = N.Index(ref)
    ix.set_hamming_weights(w)
    o = ix.make_opts(k, formula=N.FORMULA_HAMMING)
    for rep in range(2):
        ix.reset_stats()
        ix.kneighbors_device(qd.data_ptr(), nq, o, dd.data_ptr(), di.data_ptr())
        torch.cuda.synchronize()
        st = ix.stats()
    ms = st["total_kernel_ms"]
    ok = bool(np.array_equal(di[:n_chk].cpu().numpy(), oi) and np.array_equal(dd[:n_chk].cpu().numpy(), od))
    print(f"{'integer pre-filter' if mode == '1' else 'float64 scan      '}: {n_ref} refs x {nq} queries x {t} trees, k={k}, {levels} ids/tree, "
          f"{'uniform' if uniform else 'real'} weights: {ms:8.2f} ms = {n_ref * nq * t / ms / 1e9:8.2f} e12 compares/s, {nq / ms / 1e3:.3f} Mq/s; "
          f"fallbacks {st['exact_fallbacks']}; oracle slice ({n_chk} rows) bit-equal: {ok}", flush=True)
    ix.close()
