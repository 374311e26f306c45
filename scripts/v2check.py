#!/usr/bin/env python
"""Development aid: quick parity check of a library build against the oracle on a few shapes."""
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

for (n_ref, nq, d, k) in ((50000, 20000, 32, 5), (8192, 5000, 16, 5), (50000, 20000, 32, 7), (20000, 5000, 20, 3)):
    x_ref, y, x_q = synth.make_problem(n_ref, nq, d, t=2, n_dup_queries=8)
    ix = N.Index(x_ref, y)
    dist, idx = ix.kneighbors_host(x_q, ix.make_opts(k))
    od, oi = O.kneighbors(x_ref, x_q, k, "expanded")
    print(n_ref, nq, d, k, "idx equal", np.array_equal(idx, oi), "dist equal", np.array_equal(dist, od),
          "fallbacks", ix.stats()["exact_fallbacks"], flush=True)
    ix.close()
