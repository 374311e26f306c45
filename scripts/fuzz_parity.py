#!/usr/bin/env python
"""Randomised differential test against the oracle (GPU box): random shapes, neighbour counts,
duplicate / integer-valued data, both distance formulas, X=None, both ordering modes, affine maps.
usage: python scripts/fuzz_parity.py [seconds] [seed]   -- exits non-zero on the first mismatch."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = "--big" in sys.argv  # benchmark-sized reference sets (the oracle takes seconds per case)
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_cases = n_rows = 0
t_report = time.time()
while time.time() < t_end:
    d = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 24, 32, 33, 48, 64, 80, 100, 128]))
    n_ref = int(rng.choice([1, 5, 31, 32, 33, 100, 257, 1000, 3000, 7000]))
    nq = int(rng.choice([1, 31, 100, 1000, 3000]))
    if big:
        n_ref = int(rng.choice([20000, 50000, 100003]))
        nq = int(rng.choice([5000, 20000, 50001]))
        d = int(rng.choice([8, 16, 32, 64]))
        if rng.integers(0, 5) == 0:
            # more than one round of pre-filter workgroups plus a thin one: the fork / join of the finaliser on the
            # handle's side stream
            nq, n_ref = int(rng.choice([270_000, 300_001])), 20000
    kmax = min(n_ref, 34)
    k = int(rng.integers(1, kmax + 1))
    if not big and rng.integers(0, 6) == 0:
        # RFNN / GBNN: weighted Hamming distance over node-id columns (few distinct ids: exact ties are the norm),
        # uniform or random weights, both orderings, X=None; wide matrices take the column-chunked sweep
        t = int(rng.choice([1, 7, 64, 300, 1100]))
        ids_ref = rng.integers(0, int(rng.choice([2, 5, 40])), (max(n_ref, 2), t)).astype(np.float64)[:n_ref]
        ids_q = rng.integers(0, 40, (nq, t)).astype(np.float64)
        ids_q[: nq // 2] = ids_ref[rng.integers(0, n_ref, nq // 2)]  # half of the queries sit on reference rows
        w = np.full(t, 1.0 / t) if rng.integers(0, 2) else rng.random(t) + 0.01
        det = bool(rng.integers(0, 2))
        hx = N.Index(ids_ref)
        try:
            hx.set_hamming_weights(w)
            if rng.integers(0, 4) == 0 and k < n_ref:
                dist, idx = hx.kneighbors_host(None, hx.make_opts(k, exclude_self=True, deterministic=det,
                                                                  formula=N.FORMULA_HAMMING), nq=n_ref)
                od, oi = O.kneighbors_hamming(ids_ref, None, w, k, deterministic=det)
            else:
                off = int(rng.choice([0, 17]))
                dist, idx = hx.kneighbors_host(ids_q, hx.make_opts(k, deterministic=det, formula=N.FORMULA_HAMMING,
                                                                   row_offset=off))
                od, oi = O.kneighbors_hamming(ids_ref, ids_q, w, k, deterministic=det, row_offset=off)
            if not (np.array_equal(idx, oi) and np.array_equal(dist, od)):
                print(f"HAMMING MISMATCH n_ref={n_ref} nq={nq} trees={t} k={k} det={det} seed={seed}")
                sys.exit(1)
        finally:
            hx.close()
        n_cases += 1
        n_rows += nq
        continue
    kind = rng.choice(["smooth", "dup", "integer", "tiny_scale", "huge_offset", "huge_offset", "far_queries"])
    x_ref, _, x_q = synth.make_problem(max(n_ref, 2), nq, d, t=1, n_dup_refs=min(n_ref // 3, 40) if kind == "dup" else 0,
                                       n_dup_queries=min(nq // 3, 30, max(n_ref, 2) // 2) if kind == "dup" else 0)
    x_ref = x_ref[:n_ref]
    y = rng.standard_normal((n_ref, 3))
    if kind == "integer":
        x_ref, x_q = np.round(x_ref * 2.0), np.round(x_q * 2.0)
    elif kind == "tiny_scale":
        x_ref, x_q = x_ref * 1e-5, x_q * 1e-5
    elif kind == "huge_offset":
        # uncentred data: at 1e6 and beyond the reference formula's cancellation noise decides near ties
        off = float(rng.choice([1e4, 1e5, 1e6, 1e7, 1e9]))
        x_ref, x_q = x_ref + off, x_q + off
    elif kind == "far_queries":
        # some query values beyond the f16 image of the pre-filter (marked rows, exact scan)
        x_q = x_q * np.where(rng.random((nq, 1)) < 0.2, 1e7, 1.0)
    x_ref_raw = x_ref
    formula = int(rng.integers(0, 2))
    fname = "expanded" if formula == 0 else "direct"
    det = bool(rng.integers(0, 2))
    self_query = bool(rng.integers(0, 4) == 0) and k < n_ref
    row_offset = int(rng.choice([0, 0, 17, 123456]))
    # one case in three: an affine map (centre / scale / projector, d_in != d_t) applied to the
    # references on the host entry point and fused into the query preparation kernel
    affine = None
    if rng.integers(0, 3) == 0 and not self_query:
        d_t = int(rng.choice([1, 3, 8, 16, 20, 32, 40, 64]))
        c = rng.standard_normal(d) if rng.integers(0, 2) else None
        sc = (0.5 + rng.random(d)) if rng.integers(0, 2) else None
        p = rng.standard_normal((d, d_t)) if rng.integers(0, 4) else None
        affine = (c, sc, p)
        x_raw_q = x_q
        x_ref = N.affine_transform_host(x_ref, c, sc, p)
        if not np.array_equal(x_ref, O.affine(x_ref_raw, c, sc, p)):
            print(f"AFFINE MISMATCH d={d} d_t={d_t} seed={seed}")
            sys.exit(1)
        x_q = O.affine(x_raw_q, c, sc, p)
    ix = N.Index(x_ref, y)
    try:
        if affine is not None:
            ix.set_affine(d, *affine)
        if self_query:
            o = ix.make_opts(k, exclude_self=True, deterministic=det, formula=formula)
            dist, idx = ix.kneighbors_host(None, o, nq=n_ref)
            od, oi = O.kneighbors(x_ref, None, k, fname, deterministic=det)
        else:
            o = ix.make_opts(k, deterministic=det, formula=formula, row_offset=row_offset,
                             apply_affine=affine is not None)
            q_in = x_raw_q if affine is not None else x_q
            if rng.integers(0, 2):
                dist, idx = ix.kneighbors_host(q_in, o)
            else:  # device-resident tensors in and out
                qd = torch.as_tensor(np.ascontiguousarray(q_in), device="cuda")
                dd = torch.empty((nq, k), dtype=torch.float64, device="cuda")
                di = torch.empty((nq, k), dtype=torch.int64, device="cuda")
                ix.kneighbors_device(qd.data_ptr(), nq, o, dd.data_ptr(), di.data_ptr())
                torch.cuda.synchronize()
                dist, idx = dd.cpu().numpy(), di.cpu().numpy()
            od, oi = O.kneighbors(x_ref, x_q, k, fname, deterministic=det, row_offset=row_offset)
        ok = np.array_equal(idx, oi) and np.array_equal(dist, od)
        if not ok:
            bad = np.where((idx != oi).any(axis=1) | (dist != od).any(axis=1))[0]
            print(f"MISMATCH d={d} n_ref={n_ref} nq={nq} k={k} kind={kind} formula={fname} det={det} "
                  f"self={self_query} row_offset={row_offset} affine={affine is not None} seed={seed}: {len(bad)} rows, first {bad[:5]}")
            r = bad[0]
            print(" got ", idx[r], dist[r])
            print(" want", oi[r], od[r])
            sys.exit(1)
        pw = ["uniform", "distance"][int(rng.integers(0, 2))]
        if not self_query:
            pred = ix.predict_host(q_in, ix.make_opts(k, deterministic=det, formula=formula, row_offset=row_offset,
                                                      weight_mode=0 if pw == "uniform" else 1,
                                                      apply_affine=affine is not None))
            want = O.predict(y, od, oi, pw)
            if not np.allclose(pred, want, rtol=1e-12, atol=0):
                print(f"PREDICT MISMATCH d={d} n_ref={n_ref} nq={nq} k={k} kind={kind} weights={pw} seed={seed}")
                sys.exit(1)
    finally:
        ix.close()
    n_cases += 1
    n_rows += n_ref if self_query else nq
    if big or time.time() - t_report > 30:
        print(f"... {n_cases} cases, {n_rows} query rows", flush=True)
        t_report = time.time()
print(f"fuzz ok: {n_cases} cases, {n_rows} query rows, seed {seed}, {budget:.0f} s")
