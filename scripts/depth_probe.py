#!/usr/bin/env python
"""Development aid: pre-filter time as a function of the number of reference rows (fixed query batch).
The difference between consecutive sizes is the marginal cost of the tiles at that depth of the sweep:
early tiles are visited by every q-block and take many hits, late ones are mostly skipped."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = 32
g = torch.Generator(device="cuda").manual_seed(1)
xq = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(synth.mixing_matrix(d), device="cuda")
dist = torch.empty((nq, k), dtype=torch.float64, device="cuda")
idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
x_all = synth.make_features(51200, d, seed=0)
prev = (0, 0.0)
for n_ref in (256, 512, 1024, 2048, 4096, 8192, 12800, 25600, 51200):
    ix = N.Index(x_all[:n_ref])
    o = ix.make_opts(k)
    for _ in range(2):
        ix.reset_stats()
        ix.kneighbors_device(xq.data_ptr(), nq, o, dist.data_ptr(), idx.data_ptr())
        torch.cuda.synchronize()
        st = ix.stats()
    ms = st["total_coarse_ms"]
    tiles, dms = (n_ref - prev[0]) / 32, ms - prev[1]
    qblocks_per_simd = nq / 32 / 1024
    print(f"n_ref {n_ref:6d}: pre-filter {ms:7.3f} ms; marginal {dms / tiles * 1e6 / qblocks_per_simd:8.1f} ns per tile.q-block "
          f"(= {dms / tiles * 1e6 / qblocks_per_simd * 2.1:7.0f} cycles at 2.1 GHz); fallbacks {st['exact_fallbacks']}", flush=True)
    prev = (n_ref, ms)
    ix.close()
