#!/usr/bin/env python
"""Times the host-buffer (numpy in / numpy out) entry points at the bench shape.
Usage on the GPU box:  [SKNNR_HOST_CHUNK_ROWS=n] python scripts/host_path_probe.py [rows]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_ref, d, k, t = 50_000, 32, 5, 8
x_ref, y, _ = synth.make_problem(n_ref, 16, d, t=t)
g = torch.Generator(device="cuda").manual_seed(1)
xq = (torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g)
      @ torch.tensor(synth.mixing_matrix(d), device="cuda")).cpu().numpy()
ix = N.Index(x_ref, y)
o = ix.make_opts(k)
for rep in range(3):
    t0 = time.perf_counter()
    dist, idx = ix.kneighbors_host(xq, o)
    dt = time.perf_counter() - t0
    print(f"kneighbors_host {nq} rows: {dt * 1e3:.1f} ms -> {nq / dt / 1e6:.2f} Mq/s", flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    pred = ix.predict_host(xq, o)
    dt = time.perf_counter() - t0
    print(f"predict_host    {nq} rows: {dt * 1e3:.1f} ms -> {nq / dt / 1e6:.2f} Mq/s", flush=True)
print("chunk rows:", os.environ.get("SKNNR_HOST_CHUNK_ROWS", "default"))
