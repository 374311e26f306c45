#!/usr/bin/env python
"""Development aid: parity at reference sets of 0.4M and 1M rows (GPU box): python scripts/big_ref_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from sknnr_amd import _native as N
from sknnr_amd import synth
for n_ref, nq, d, k in ((1_000_003, 20_000, 8, 5), (400_001, 10_000, 40, 3)):
    x_ref = synth.make_features(n_ref, d, seed=0)
    x_q = synth.make_features(nq, d, seed=1)
    t0 = time.time()
    ix = N.Index(x_ref)
    dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k))
    t1 = time.time()
    od, oi = O.kneighbors(x_ref, x_q, k, "expanded")
    print(f"n_ref={n_ref} nq={nq} d={d} k={k}: gpu {t1 - t0:.2f} s (incl. index build), oracle {time.time() - t1:.1f} s, "
          f"idx equal {np.array_equal(idx, oi)}, dist equal {np.array_equal(dist, od)}, stats {ix.stats()}", flush=True)
    ix.close()
