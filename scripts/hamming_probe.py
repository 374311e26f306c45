#!/usr/bin/env python
"""Development aid: the weighted-Hamming search on the device against the reference's CPU path (scikit-learn
KNeighborsRegressor(algorithm="brute", metric="hamming", metric_params={"w": w}) = scipy cdist + argpartition) on
random node-id matrices.  usage: hamming_probe.py n_ref nq n_trees k"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402

n_ref, nq, t, k = (int(a) for a in sys.argv[1:5])
rng = np.random.default_rng(0)
ref = rng.integers(0, 40, (n_ref, t)).astype(np.float64)
q = rng.integers(0, 40, (nq, t)).astype(np.float64)
w = rng.random(t) + 0.01
ix = N.Index(ref)
ix.set_hamming_weights(w)
o = ix.make_opts(k, formula=N.FORMULA_HAMMING)
ix.kneighbors_host(q[:1000], o)
t0 = time.perf_counter()
dist, idx = ix.kneighbors_host(q, o)
gpu = time.perf_counter() - t0
st = ix.stats()
from sklearn.neighbors import KNeighborsRegressor

reg = KNeighborsRegressor(n_neighbors=k, algorithm="brute", metric="hamming", metric_params={"w": w}).fit(ref, np.zeros(n_ref))
n_cpu = min(nq, 2000)
t0 = time.perf_counter()
cd, ci = reg.kneighbors(q[:n_cpu])
cpu = (time.perf_counter() - t0) * nq / n_cpu
print(f"{n_ref} refs x {nq} queries x {t} trees, k={k}: device {gpu * 1e3:.1f} ms wall ({st['last_kernel_ms']:.1f} ms kernels) = "
      f"{nq / gpu / 1e6:.3f} Mq/s; sklearn/scipy CPU ({os.cpu_count()} cpus, {n_cpu}-row sample scaled) {cpu * 1e3:.0f} ms = {nq / cpu / 1e6:.4f} Mq/s; "
      f"sorted distances equal on the sample: {np.array_equal(np.sort(dist[:n_cpu], 1), np.sort(cd, 1))}")
