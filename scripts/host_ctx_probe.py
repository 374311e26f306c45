#!/usr/bin/env python
"""Development aid: the one-shot host call in bench.py's setting (engine with the CCA affine map, 10M x 32 rows), call by
call, before and after device-resident 10M-row calls, with and without the affine map."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

nq, k = 10_000_000, 5
eng, x_ref_t, affine, y, _ = bench.fit_space("gnn", 50_000, 32, 40, 0)
q = bench.gen_queries(nq, 32, 1234, torch)
q_host = q.cpu().numpy()


def host_calls(tag, x, n, **kw):
    for rep in range(n):
        t0 = time.perf_counter()
        d, i = eng.kneighbors(x, k, **kw)
        dt = time.perf_counter() - t0
        print(f"{tag} call {rep}: {dt * 1e3:.1f} ms -> {nq / dt / 1e6:.1f} Mq/s", flush=True)
        del d, i


mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode in ("all", "host_first"):
    host_calls("affine, before any device call", q_host, 4, apply_affine=True)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.kneighbors(q, k, apply_affine=True)
    torch.cuda.synchronize()
    print(f"device-resident 10M call: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
host_calls("affine, after 10M device calls", q_host, 3, apply_affine=True)
for share in (5_000_000, 2_500_000, 1_250_000):
    for rep in range(2):
        eng.kneighbors(q[:share], k, apply_affine=True)
    torch.cuda.synchronize()
    host_calls(f"affine, after {share}-row device calls", q_host, 2, apply_affine=True)
c, s, p = affine
from sknnr_amd._native import affine_transform_host  # noqa: E402
qt = affine_transform_host(q_host[:], c, s, p, device=0)
host_calls("pre-transformed rows, no affine", np.ascontiguousarray(qt), 3)
