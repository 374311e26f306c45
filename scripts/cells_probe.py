#!/usr/bin/env python
"""Development aid (round 3): query bucketing / cell-ordered image A/B on the benchmark workload.
SKNNR_CELLS (read when an index is built): 0 = the round-2 order (no bucketing), 2..6 = tree depth.
Prints pre-filter ms, all-kernel ms, exact fallbacks and a slice checked against the oracle."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
depths = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 6]
kind = sys.argv[4] if len(sys.argv) > 4 else "gnn"
d_in = int(sys.argv[5]) if len(sys.argv) > 5 else 32
q = bench.gen_queries(rows, d_in, 1000, torch)
for depth in depths:
    os.environ["SKNNR_CELLS"] = str(depth)
    eng, x_ref_t, affine, y, _ = bench.fit_space(kind, 50_000, d_in, 40, 0)
    for rep in range(3):
        eng.reset_stats()
        t0 = time.perf_counter()
        dist, idx = eng.kneighbors(q, k, apply_affine=True)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        st = eng.stats()
    n_chk = 100_000
    chk = bench.oracle_slice_check(x_ref_t, affine, q[:n_chk].cpu().numpy(), k, dist[:n_chk].cpu().numpy(), idx[:n_chk].cpu().numpy())
    tf, frac = bench.mfma_frac(st["coarse_rows_timed"], 50_000, x_ref_t.shape[1], st["total_coarse_ms"])
    print(f"cells depth {depth}: rows {rows} k {k} {kind} d={x_ref_t.shape[1]}: pre-filter {st['total_coarse_ms']:.2f} ms (frac {frac:.3f}), all kernels "
          f"{st['total_kernel_ms']:.2f} ms, wall {wall * 1e3:.2f} ms = {rows / wall / 1e6:.1f} Mq/s, fallbacks {st['exact_fallbacks']}, oracle {chk}", flush=True)
    eng.close()
