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
# the same rows through a tile stream into preallocated, already touched output arrays (no allocation or
# first-touch page faults inside the timed region: what a raster job writing into memmaps / reused buffers sees)
d_out = np.zeros((nq, k))
i_out = np.zeros((nq, k), dtype=np.int64)
tile = 1_000_000
for rep in range(3):
    t0 = time.perf_counter()
    with ix.open_stream(o) as s:
        for a in range(0, nq, tile):
            s.push(xq[a:a + tile], out_idx=i_out[a:a + tile], out_dist=d_out[a:a + tile])
    dt = time.perf_counter() - t0
    print(f"stream of {tile}-row pushes, reused outputs: {dt * 1e3:.1f} ms -> {nq / dt / 1e6:.2f} Mq/s", flush=True)
print("same as the one-shot call:", bool(np.array_equal(i_out, idx) and np.array_equal(d_out, dist)))
