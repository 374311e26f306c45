#!/usr/bin/env python
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one ``kneighbors`` pass of the transformed estimator path over one batch of
synthetic query rows already resident in HBM:
    affine transform (32 -> 32, the GNN/CCA form)  ->  f16x3 MFMA pre-filter over all
    references  ->  float64 re-score + certificate + sknnr reorder  ->  (dist, idx) in HBM,
and, with N > 1, the RCCL all-gather of the per-rank (dist, idx) blocks (weak scaling: every
rank answers its own ``--rows`` query rows against the replicated reference set).

Rank 0 prints ONE JSON line: BASELINE.json's metric (Mqueries/s, whole job), plus
``roofline`` (dominant kernel = the MFMA pre-filter, timed with HIP events on its launch
stream inside the library) and ``cpu_baseline`` (the reference's CPU arithmetic -- sklearn's
ArgKmin + the restated reorder -- timed on this host on a bounded sample; N=1 only).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=10_000_000, help="query rows per GPU")
    ap.add_argument("--refs", type=int, default=50_000)
    ap.add_argument("--dims", type=int, default=32)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--targets", type=int, default=40)
    ap.add_argument("--cpu-sample", type=int, default=400_000, help="rows of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather of (dist, idx)")
    ap.add_argument("--predict", action="store_true", help="also time predict (distance weights) as an extra")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the all-gather even with one rank (self-test of the N>1 path)")
    return ap.parse_args()


def make_queries(rows, dims, rank, torch):
    """Synthetic law of SURVEY.md 8(d), generated on the device: Z @ (I + 0.3 G)."""
    from sknnr_amd import synth

    g = torch.Generator(device="cuda").manual_seed(1000 + rank)
    mix = torch.tensor(synth.mixing_matrix(dims), device="cuda")
    out = torch.empty((rows, dims), dtype=torch.float64, device="cuda")
    step = 1 << 21
    for a in range(0, rows, step):
        b = min(rows, a + step)
        out[a:b] = torch.randn((b - a, dims), dtype=torch.float64, device="cuda", generator=g) @ mix
    return out


def cpu_baseline(x_ref_t, center, proj, q_raw_sample, k):
    """The reference's CPU path on this host: numpy transform + sklearn ArgKmin (what
    RawKNNRegressor.kneighbors calls, REF _base.py:162-164) + the restated reorder."""
    from oracle import oracle as O

    info = {"unit": "Mqueries/s", "sample": f"{len(q_raw_sample)} query rows of the same workload (prefix of rank 0's batch)"}
    try:
        import sklearn
        from sklearn.neighbors import KNeighborsRegressor
        from threadpoolctl import threadpool_info

        reg = KNeighborsRegressor(n_neighbors=k, algorithm="auto").fit(x_ref_t, np.zeros(len(x_ref_t)))
        t0 = time.perf_counter()
        q_t = (q_raw_sample - center) @ proj
        dist, idx = reg.kneighbors(q_t)
        dist, idx = O.deterministic_reorder(dist, idx)
        dt = time.perf_counter() - t0
        threads = [p.get("num_threads") for p in threadpool_info() if p.get("user_api") == "openmp"]
        info.update(value=len(q_raw_sample) / dt / 1e6, cores=int(max(threads) if threads else os.cpu_count()),
                    kind="reference", engine=f"scikit-learn {sklearn.__version__} ArgKmin ({reg._fit_method}) + restated reorder",
                    host_cpus=os.cpu_count(), seconds=dt)
        return info, (dist, idx)
    except Exception as err:  # sklearn missing on the box: time the C restatement instead
        t0 = time.perf_counter()
        q_t = O.affine(q_raw_sample, center, None, proj)
        dist, idx = O.kneighbors(x_ref_t, q_t, k, "expanded")
        dt = time.perf_counter() - t0
        info.update(value=len(q_raw_sample) / dt / 1e6, cores=O.num_threads(), kind="port",
                    engine=f"oracle/knn_oracle.c (OpenMP); sklearn unavailable: {err}", host_cpus=os.cpu_count(), seconds=dt)
        return info, (dist, idx)


def main():
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from sknnr_amd import synth
    from sknnr_amd._engine import KNNEngine
    from sknnr_amd._native import affine_transform_host
    from sknnr_amd.distributed import cyclic_slot
    from sknnr_amd.transformers import CCATransformer

    # ---- fit (host, once): CCA ordination of the synthetic reference set -> affine map ------
    t_fit = time.perf_counter()
    x_ref = synth.make_features(args.refs, args.dims, seed=0)
    y_ref = synth.make_targets(x_ref, t=args.targets, kind="positive")
    tr = CCATransformer().fit(x_ref, y_ref)
    center, _, proj = tr.affine_params()
    d_t = proj.shape[1]
    x_ref_t = affine_transform_host(x_ref, center, None, proj, device=local_rank)
    eng = KNNEngine(x_ref_t, y_ref, device=local_rank)
    eng.set_affine(args.dims, center, None, proj)
    t_fit = time.perf_counter() - t_fit

    q = make_queries(args.rows, args.dims, rank, torch)
    nq, k = args.rows, args.k
    stream = torch.cuda.current_stream()

    # N > 1: four all-gather chunks (chunk i travels while chunk i+1 is computed), cut at whole rounds
    # of the pre-filter grid (256 CUs x 1024 rows per workgroup) and shrinking, so that the last
    # gather -- the only one nothing hides -- is the smallest
    round_rows = 256 * 1024
    n_rounds = -(-args.rows // round_rows)
    cuts, acc_w = [0], 0.0
    for w_ in (0.31, 0.28, 0.23):
        acc_w += w_
        cuts.append(min(args.rows, max(cuts[-1], int(round(acc_w * n_rounds)) * round_rows)))
    cuts.append(args.rows)
    gather_chunks = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    comm_stream = torch.cuda.Stream() if use_dist else None
    gather_in_place = [True]

    def step():
        if not (use_dist and not args.no_gather):
            return eng.kneighbors(q, k, apply_affine=True, deterministic=True, row_offset=rank * nq)
        # N > 1: the job's rows are dealt to the ranks chunk-cyclically -- chunk c of every rank forms
        # the contiguous global rows [W*a, W*b) with this rank's block at slot `rank` -- so each rank
        # writes its chunk straight into its slot of the final arrays and the RCCL all-gather of a
        # finished chunk is in place (no staging copies), on a side stream, under the next chunk's
        # kernels.  (Every block is contiguous, so its global row offset is all the reorder needs.)
        d_all = torch.empty((world * nq, k), dtype=torch.float64, device="cuda")
        i_all = torch.empty((world * nq, k), dtype=torch.int64, device="cuda")
        works = []
        for a, b in gather_chunks:
            lo = cyclic_slot(world, rank, a, b)
            d_own, i_own = d_all[lo: lo + (b - a)], i_all[lo: lo + (b - a)]
            eng.kneighbors(q[a:b], k, apply_affine=True, deterministic=True, row_offset=lo, out=(d_own, i_own))
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream())
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(done)
                for whole, own in ((d_all[world * a: world * b], d_own), (i_all[world * a: world * b], i_own)):
                    try:
                        # in place: the send buffer is this rank's slot of the receive buffer
                        works.append(dist.all_gather_into_tensor(whole, own if gather_in_place[0] else own.clone(),
                                                                 async_op=True))
                    except RuntimeError:
                        if not gather_in_place[0]:
                            raise
                        gather_in_place[0] = False  # a backend that rejects aliasing: stage the slot once
                        works.append(dist.all_gather_into_tensor(whole, own.clone(), async_op=True))
        for w in works:
            w.wait()
        torch.cuda.current_stream().wait_stream(comm_stream)
        return d_all, i_all

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    coarse_ms = 0.0
    kernel_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d_out, i_out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel device time of the LAST step (HIP events recorded by the library on the launch stream)
    st = eng.stats()
    coarse_ms, kernel_ms = st["last_coarse_ms"], st["last_kernel_ms"]

    t_max = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * nq * args.steps / elapsed / 1e6

    extra = {}
    if args.predict:
        for _ in range(max(1, args.warmup)):
            p = eng.predict(q, 7, "distance", apply_affine=True)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            p = eng.predict(q, 7, "distance", apply_affine=True)
        barrier()
        extra["predict_k7_distance_Mq_s"] = nq * args.steps / (time.perf_counter() - t1) / 1e6
        del p

    if rank == 0:
        alg_flops = 2.0 * nq * args.refs * d_t  # 2 Nq Nref D_t per pass
        achieved_tf = alg_flops / (coarse_ms * 1e-3) / 1e12
        alg_bytes = nq * args.dims * 8 + args.refs * d_t * 8 + nq * k * 16
        traffic, traffic_note = None, None
        pmc_file = os.path.join(ROOT, "profiles", "r01_coarse_pmc.json")
        if os.path.exists(pmc_file) and (args.refs, d_t, k) == (50_000, 32, 5):
            # PMC counters cannot be collected inside this process; the committed rocprofv3 --pmc
            # passes of this same command give HBM bytes per query row for the dominant kernel
            # (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section), scaled to this step.
            pmc = json.load(open(pmc_file))
            traffic = pmc["hbm_bytes_per_query_row"] * nq
            traffic_note = "profiles/r01_coarse_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE), bytes per step"
        roofline = {
            "kernel": "sknnr::coarse_kernel<KS=%d,M=%d,NQB=2> (f16x3 split MFMA pre-filter, correction products skipped when no lane can hit; lane-local top-M)" % ((d_t + 15) // 16, 6 if k <= 5 else 8),
            "bound": "mfma", "achieved": achieved_tf, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved_tf / PEAK_F16_MFMA_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
            "executed_over_algorithmic_max": 3.0,
            "vs_f32_mfma_peak": achieved_tf / PEAK_F32_MFMA_TFLOPS,
            "kernel_ms_per_step": coarse_ms, "all_kernels_ms_per_step": kernel_ms,
            "hbm_algorithmic_GBs": alg_bytes / (kernel_ms * 1e-3) / 1e9,
            "hbm_frac_of_peak": alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
        }
        result = {
            "metric": "Mqueries/sec + achieved HBM GB/s, 10M x 50k x 32 k=5, 1/2/4/8 MI355X",
            "value": value, "unit": "Mqueries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3 split MFMA (f32 accumulate) pre-filter + f64 exact re-score",
            "data": "synthetic",
            "config": {
                "workload": f"GNN-style kneighbors: affine {args.dims}->{d_t} (CCA fit on synthetic refs) + "
                            f"{nq} query rows/GPU x {args.refs} refs x {d_t} dims, k={k}, deterministic reorder, "
                            "float64 (dist, idx) out" + (" + RCCL all-gather" if use_dist and not args.no_gather else ""),
                "rows_per_gpu": nq, "n_ref": args.refs, "d_in": args.dims, "d_t": int(d_t), "k": k,
                "parallelism": f"query-row shards x{world}",
            },
            "roofline": roofline,
            "fit_seconds": t_fit,
            "exact_fallbacks": int(st["exact_fallbacks"]), "queries_answered": int(st["queries"]),
            **extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            n_s = min(args.cpu_sample, nq)
            q_host = q[:n_s].cpu().numpy()
            base, (cd, ci) = cpu_baseline(x_ref_t, center, proj, q_host, k)
            result["cpu_baseline"] = base
            gi = i_out[rank * nq: rank * nq + n_s].cpu().numpy() if use_dist and not args.no_gather else i_out[:n_s].cpu().numpy()
            gd = d_out[rank * nq: rank * nq + n_s].cpu().numpy() if use_dist and not args.no_gather else d_out[:n_s].cpu().numpy()
            rel = np.abs(gd - cd) / np.maximum(np.abs(cd), 1e-300)
            result["parity_vs_cpu_reference"] = {
                "rows": int(n_s), "index_rows_equal": int((gi == ci).all(axis=1).sum()),
                "max_rel_dist_err": float(rel.max()),
            }
            result["speedup_vs_cpu"] = value / base["value"]
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
