#!/usr/bin/env python
"""Exact-scan throughput (GPU box): a call outside the MFMA envelope (n_neighbors = 33) and an integer-valued
workload where most rows tie at the k-th slot.  usage: [SKNNR_HIP_LIBRARY=variant.so] python scripts/scan_probe.py"""
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

n_ref, d = 50_000, 32
x_ref, _, x_q = synth.make_problem(n_ref, 200_000, d, t=1)
ix = N.Index(x_ref)
xq = torch.as_tensor(x_q, device="cuda")
for k, nq in ((33, 100_000), (5, 200_000)):
    xr = x_ref if k == 33 else np.round(x_ref * 2.0)
    q = xq[:nq] if k == 33 else torch.round(xq[:nq] * 2.0)
    jx = ix if k == 33 else N.Index(xr)
    o = jx.make_opts(k)
    dist = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    for rep in range(3):
        jx.reset_stats()
        jx.kneighbors_device(q.data_ptr(), nq, o, dist.data_ptr(), idx.data_ptr())
        torch.cuda.synchronize()
        st = jx.stats()
    print(f"k={k} {'exact-only' if k == 33 else 'integer-valued features'}: {nq} rows, kernels {st['last_kernel_ms']:.2f} ms "
          f"(pre-filter {st['last_coarse_ms']:.2f}) -> {nq / st['last_kernel_ms'] / 1e3:.2f} Mq/s; exact-scanned rows "
          f"{st['exact_fallbacks'] + st['exact_only_queries']}", flush=True)
print("library:", os.environ.get("SKNNR_HIP_LIBRARY", "in-tree"))
