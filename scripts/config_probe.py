#!/usr/bin/env python
"""Times the BASELINE.json configurations other than the bench line on one MI355X (device-resident
inputs), with a slice checked against the oracle.  Usage on the GPU box: python scripts/config_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402


def run(name, nq, n_ref, d, k, t=0, weight_mode=0, reps=3):
    x_ref, y, _ = synth.make_problem(n_ref, 16, d, t=max(t, 1))
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(
        synth.mixing_matrix(d), device="cuda")
    ix = N.Index(x_ref, y if t else None)
    o = ix.make_opts(k, weight_mode=weight_mode)
    dist = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    pred = torch.empty((nq, max(t, 1)), dtype=torch.float64, device="cuda")
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if t:
            ix.predict_device(q.data_ptr(), nq, o, pred.data_ptr(), dist.data_ptr(), idx.data_ptr())
        else:
            ix.kneighbors_device(q.data_ptr(), nq, o, dist.data_ptr(), idx.data_ptr())
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    s = ix.stats()
    sl = slice(0, 1024)
    od, oi = O.kneighbors(x_ref, q[sl].cpu().numpy(), k, "expanded")
    bad = int((idx[sl].cpu().numpy() != oi).any(axis=1).sum())
    print(f"{name}: {nq}x{n_ref}x{d} k={k} t={t}: {best * 1e3:.1f} ms -> {nq / best / 1e6:.1f} Mq/s "
          f"(coarse {s['last_coarse_ms']:.1f} ms of {s['last_kernel_ms']:.1f}; fallbacks/call "
          f"{s['exact_fallbacks'] / reps:.0f}); slice bad rows {bad}", flush=True)
    ix.close()


if __name__ == "__main__":
    run("configs[1] Euclidean", 1_000_000, 10_000, 16, 5)
    run("bench line", 10_000_000, 50_000, 32, 5)
    run("configs[2] GNN k=7 distance-weighted predict T=40", 10_000_000, 50_000, 32, 7, t=40, weight_mode=1)
    run("configs[3] Mahalanobis d=64 (one GPU's share of 10M/8)", 1_250_000, 50_000, 64, 5)
    run("configs[3] Mahalanobis d=64, 10M on one GPU", 10_000_000, 50_000, 64, 5, reps=2)
    run("configs[4] MSN d=8 k=1 100k refs (one GPU's share of 50M/8)", 6_250_000, 100_000, 8, 1)
    run("k=10", 2_000_000, 50_000, 32, 10, reps=2)
    run("k=20", 1_000_000, 50_000, 32, 20, reps=2)
