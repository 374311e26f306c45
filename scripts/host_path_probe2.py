#!/usr/bin/env python
"""Development aid: the one-shot host call and the tile stream under the environment's pipeline knobs
(SKNNR_HOST_CHUNK_ROWS, SKNNR_NO_PREFAULT), plus ten back-to-back device-resident 1M-row calls (the GPU-side floor of
tile-wise processing)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

nq = 10_000_000
n_ref, d, k = 50_000, 32, 5
x_ref, y, _ = synth.make_problem(n_ref, 16, d, t=8)
g = torch.Generator(device="cuda").manual_seed(1)
xq_dev = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(synth.mixing_matrix(d), device="cuda")
xq = xq_dev.cpu().numpy()
ix = N.Index(x_ref, y)
o = ix.make_opts(k)
tile = int(os.environ.get("SKNNR_HOST_CHUNK_ROWS", 1_000_000))
dd = torch.empty((nq, k), dtype=torch.float64, device="cuda")
di = torch.empty((nq, k), dtype=torch.int64, device="cuda")
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a in range(0, nq, tile):
        n = min(tile, nq - a)
        ix.kneighbors_device(xq_dev[a:].data_ptr(), n, ix.make_opts(k, row_offset=a), dd[a:].data_ptr(), di[a:].data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"tile {tile}: {nq // tile} device-resident calls back to back: {dt * 1e3:.1f} ms")
best = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    dist, idx = ix.kneighbors_host(xq, o)
    dt = time.perf_counter() - t0
    best = min(best, dt)
    del dist, idx
print(f"tile {tile}: one-shot kneighbors_host best of 4: {best * 1e3:.1f} ms -> {nq / best / 1e6:.1f} Mq/s")
d_out = np.zeros((nq, k))
i_out = np.zeros((nq, k), dtype=np.int64)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    with ix.open_stream(o) as s:
        for a in range(0, nq, 1_000_000):
            s.push(xq[a:a + 1_000_000], out_idx=i_out[a:a + 1_000_000], out_dist=d_out[a:a + 1_000_000])
    best = min(best, time.perf_counter() - t0)
print(f"tile {tile}: stream of 1M-row pushes, reused outputs: {best * 1e3:.1f} ms -> {nq / best / 1e6:.1f} Mq/s")
t0 = time.perf_counter()
d2, i2 = ix.kneighbors_host(xq, o)
print("one-shot into fresh arrays incl. allocation:", f"{(time.perf_counter() - t0) * 1e3:.1f} ms", "equal:", bool(np.array_equal(i2, i_out) and np.array_equal(d2, d_out)))
