#!/usr/bin/env python
"""Development aid: the benchmark shape with spatially coherent queries (consecutive rows similar, as
the pixels of a raster scanned in order are) against the iid queries BASELINE prescribes.
usage (GPU box): python scripts/coherent_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

nq, n_ref, d, k = 10_000_000, 50_000, 32, 5
x_ref, _, _ = synth.make_problem(n_ref, 16, d, t=1)
mix = torch.tensor(synth.mixing_matrix(d), device="cuda")
g = torch.Generator(device="cuda").manual_seed(7)
ix = N.Index(x_ref)
o = ix.make_opts(k)
dist = torch.empty((nq, k), dtype=torch.float64, device="cuda")
idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
for name, rho in (("iid rows", 0.0), ("AR(1) rows, rho 0.9", 0.9), ("AR(1) rows, rho 0.99", 0.99), ("AR(1) rows, rho 0.999", 0.999)):
    z = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g)
    if rho > 0:
        # stationary AR(1) along the rows, in blocks of 4096 rows (block starts are independent draws)
        z = z.view(-1, 4096, d) if nq % 4096 == 0 else z[: nq - nq % 4096].view(-1, 4096, d)
        s = (1.0 - rho * rho) ** 0.5
        out = torch.empty_like(z)
        prev = z[:, 0]
        out[:, 0] = prev
        for i in range(1, 4096):
            prev = rho * prev + s * z[:, i]
            out[:, i] = prev
        z = out.reshape(-1, d)
    q = (z @ mix).contiguous()
    n = q.shape[0]
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.kneighbors_device(q.data_ptr(), n, o, dist.data_ptr(), idx.data_ptr())
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    st = ix.stats()
    print(f"{name}: {n} x {n_ref} x {d} k={k}: {best * 1e3:.1f} ms -> {n / best / 1e6:.1f} Mq/s "
          f"(pre-filter {st['last_coarse_ms']:.1f} ms)", flush=True)
