#!/usr/bin/env python
"""Exploratory GPU probe (development aid, not a test): validates the MFMA operand maps,
measures the split-contraction error, checks kneighbors/predict against the oracle and
times a few shapes.  Usage on the GPU box:  python scripts/gpu_probe.py [--big]"""

from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (before the HIP library: see sknnr_amd/_native.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402


def coarse_error(d, n_ref=2048, nq=256, scale_rows=1.0):
    x_ref, _, x_q = synth.make_problem(n_ref, nq, d, t=2)
    x_ref = x_ref * scale_rows
    x_q = x_q * scale_rows
    ix = N.Index(x_ref)
    m, qn, s, eps = ix.debug_coarse_matrix(x_q)
    mu = x_ref.mean(axis=0)
    rp = s * (x_ref - mu)
    qp = s * (x_q - mu)
    exact = (rp * rp).sum(1)[None, :] - 2.0 * qp @ rp.T
    err = np.abs(m.astype(np.float64) - exact)
    unit = 2.0 ** -24 * (np.sqrt((qp * qp).sum(1))[:, None] + np.sqrt((rp * rp).sum(1))[None, :]) ** 2
    qn_err = np.abs(qn - (qp * qp).sum(1)).max()
    print(f"coarse d={d:3d} scale={scale_rows:g}: s={s:g} max|err|={err.max():.3e} "
          f"max err/unit={np.max(err / unit):.3f} mean={np.mean(err / unit):.4f} qn_err={qn_err:.2e} "
          f"corr={np.corrcoef(m.ravel(), exact.ravel())[0, 1]:.9f}")
    ix.close()


def parity(d, n_ref=2048, nq=1024, k=5, dup=False):
    x_ref, y, x_q = synth.make_problem(n_ref, nq, d, t=6, n_dup_refs=96 if dup else 0,
                                       n_dup_queries=64 if dup else 0)
    ix = N.Index(x_ref, y)
    for formula, fname in ((0, "expanded"), (1, "direct")):
        o = ix.make_opts(k, formula=formula)
        dist, idx = ix.kneighbors_host(x_q, o)
        od, oi = O.kneighbors(x_ref, x_q, k, fname)
        bad = int((idx != oi).any(axis=1).sum())
        derr = float(np.abs(dist - od).max())
        o = ix.make_opts(k, formula=formula, exclude_self=True)
        dist, idx = ix.kneighbors_host(None, o, nq=n_ref)
        od, oi = O.kneighbors(x_ref, None, k, fname)
        bad_s = int((idx != oi).any(axis=1).sum())
        derr_s = float(np.abs(dist - od).max())
        print(f"parity d={d} k={k} dup={dup} {fname}: tgt bad rows {bad} derr {derr:.2e} | "
              f"self bad rows {bad_s} derr {derr_s:.2e} | stats {ix.stats()}")
    for mode, w in ((0, "uniform"), (1, "distance")):
        o = ix.make_opts(k, weight_mode=mode)
        pred = ix.predict_host(x_q, o)
        od, oi = O.kneighbors(x_ref, x_q, k, "expanded")
        op = O.predict(y, od, oi, w)
        print(f"   predict {w}: max err {np.abs(pred - op).max():.2e}")
    ix.close()


def timing(nq, n_ref, d, k, reps=3):
    import torch

    x_ref = synth.make_features(n_ref, d, seed=0)
    ix = N.Index(x_ref)
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(
        synth.mixing_matrix(d), device="cuda")
    dist = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    o = ix.make_opts(k)
    st = torch.cuda.current_stream().cuda_stream
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.kneighbors_device(q.data_ptr(), nq, o, dist.data_ptr(), idx.data_ptr(), st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s = ix.stats()
        print(f"timing {nq}x{n_ref}x{d} k={k}: {dt * 1e3:.1f} ms wall, kernel {s['last_kernel_ms']:.1f} ms, "
              f"coarse {s['last_coarse_ms']:.1f} ms -> {nq / dt / 1e6:.2f} Mq/s, "
              f"coarse TF(alg) {2.0 * nq * n_ref * d / (s['last_coarse_ms'] * 1e-3) / 1e12:.1f}, "
              f"fallbacks {s['exact_fallbacks']}")
    # host-buffer entry point (PCIe staging included): numpy in, numpy out
    n_h = min(nq, 2_000_000)
    qh = q[:n_h].cpu().numpy()
    t0 = time.perf_counter()
    ix.kneighbors_host(qh, o)
    dt = time.perf_counter() - t0
    print(f"   host buffers (PCIe inclusive), {n_h} rows: {dt * 1e3:.1f} ms -> {n_h / dt / 1e6:.2f} Mq/s")
    # spot parity on a slice
    sl = slice(0, 2048)
    od, oi = O.kneighbors(x_ref, q[sl].cpu().numpy(), k, "expanded")
    print("   slice parity bad rows:", int((idx[sl].cpu().numpy() != oi).any(axis=1).sum()),
          "derr", float(np.abs(dist[sl].cpu().numpy() - od).max()))
    ix.close()


if __name__ == "__main__":
    print("devices:", N.device_count())
    for d in (8, 16, 32, 64, 100):
        coarse_error(d)
    coarse_error(32, scale_rows=1e-3)
    coarse_error(32, scale_rows=1e4)
    for d in (8, 16, 32, 64):
        parity(d)
    parity(32, dup=True)
    parity(16, k=7, dup=True)
    parity(32, k=1)
    parity(20, k=9)   # k outside the MFMA list -> exact scan only
    timing(1 << 20, 10000, 16, 5)
    timing(1 << 21, 50000, 32, 5)
    if "--big" in sys.argv:
        timing(10_000_000, 50000, 32, 5)
