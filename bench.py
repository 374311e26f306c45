#!/usr/bin/env python
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one ``kneighbors`` pass of the transformed estimator path over one batch of
synthetic query rows already resident in HBM:
    affine transform (32 -> 32, the GNN/CCA form)  ->  split-f16 MFMA pre-filter over all
    references  ->  float64 re-score + certificate + sknnr reorder  ->  (dist, idx) in HBM,
and, with N > 1, the RCCL all-gather of the per-rank (dist, idx) blocks.

N = 1 is BASELINE.json's headline shape (10M x 50k x 32, k = 5).  N > 1 defaults to STRONG
scaling -- the same 10M-row job split over the ranks (`--scaling weak`: 10M rows per rank) --
because BASELINE.json's configurations shard a fixed job over the GPUs of a node.

Rank 0 prints ONE JSON line: BASELINE.json's metric (Mqueries/s, whole job) with
  ``roofline``      dominant kernel = the MFMA pre-filter, HIP events on its launch stream inside
                    the library, summed over the timed region;
  ``cpu_baseline``  the reference's CPU arithmetic (sklearn ArgKmin + restated reorder) on this host
                    on a bounded sample (N = 1 only);
and, at N = 1 unless ``--no-extras``, ``configs`` (BASELINE.json configs 2-5 at one GPU, each with an
oracle-checked slice), ``host_to_host_Mq_s`` / ``stream_Mq_s`` / ``estimator_Mq_s`` (the same
workload entered from numpy arrays: PCIe-inclusive, never ``value``), ``cpu_baseline_c5`` (kd_tree
and brute) and ``laws`` (reference/query laws that are hostile to the tile-level skip test).
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30, help="timed passes (default: ~1.5 s of device time, enough for a 1 Hz SMI sampler to see it)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000,
                    help="query rows of the job (strong scaling: in total; weak: per GPU)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--refs", type=int, default=50_000)
    ap.add_argument("--dims", type=int, default=32)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--targets", type=int, default=40)
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="rows of the CPU baseline sample (about 12 s on the GPU box's host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (skip configs 2-5, host paths, laws)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather of (dist, idx)")
    ap.add_argument("--gather-chunks", type=int, default=2, choices=[1, 2, 3, 4], help="N>1: calls a rank's share is split into (all but the last gather overlap compute)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the all-gather even with one rank (self-test of the N>1 path)")
    ap.add_argument("--traffic", choices=["auto", "measure", "committed", "off"], default="auto",
                    help="HBM bytes of the dominant kernel and of the step: 'measure' runs this bench twice more as a CHILD under "
                         "`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (counters only, no trace domains) before this process "
                         "touches the GPU; 'auto' does so at N = 1 when rocprofv3 is on PATH, else quotes the committed passes")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1: 'nccl' = RCCL over xGMI, one rank per GPU (the measured configuration); 'gloo' rehearses the "
                         "same step with several ranks on ONE GPU (RCCL refuses that), collectives on CUDA tensors over gloo")
    ap.add_argument("--dry-launch", action="store_true",
                    help="--gpus N > 1 outside a launcher: print the command that would start the N ranks and exit")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch: rendezvous port (default: a free one)")
    return ap.parse_args(argv)


def launch_command(argv, n_gpus, port):
    """The command `python bench.py --gpus N` (N > 1, no launcher around it) runs as its child: torch.distributed.run with
    one rank per GPU on this node, rendezvous on 127.0.0.1, the same bench arguments."""
    passed = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *passed]


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks as CHILD processes
    (this parent has not touched the GPU -- torch is not even imported yet -- and never re-execs), relay rank 0's JSON
    line (the children inherit stdout) and exit with the children's status."""
    import socket
    import subprocess

    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = launch_command(argv, args.gpus, port)
    if args.dry_launch:
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


def gen_queries(rows, dims, seed, torch, law="baseline"):
    """Synthetic query rows generated on the device.  'baseline': SURVEY.md 8(d), Z @ (I + 0.3 G)."""
    from sknnr_amd import synth

    g = torch.Generator(device="cuda").manual_seed(seed)
    out = torch.empty((rows, dims), dtype=torch.float64, device="cuda")
    step = 1 << 21
    mix = torch.tensor(synth.mixing_matrix(dims), device="cuda") if law == "baseline" else None
    for a in range(0, rows, step):
        b = min(rows, a + step)
        if law == "baseline":
            out[a:b] = torch.randn((b - a, dims), dtype=torch.float64, device="cuda", generator=g) @ mix
        else:  # uniform hypercube
            out[a:b] = torch.rand((b - a, dims), dtype=torch.float64, device="cuda", generator=g)
    return out


def fit_space(kind, n_ref, d_in, t, device, n_components=None, x_ref=None):
    """Fit the feature space of one estimator family on synthetic references and build the engine:
    returns (engine, x_ref_t, (center, scale, proj), y, fit_seconds)."""
    from sknnr_amd import synth
    from sknnr_amd import transformers as T
    from sknnr_amd._engine import KNNEngine
    from sknnr_amd._native import affine_transform_host

    t0 = time.perf_counter()
    if x_ref is None:
        x_ref = synth.make_features(n_ref, d_in, seed=0)
    y = synth.make_targets(x_ref, t=t, kind="positive" if kind == "gnn" else "linear")
    if kind == "gnn":
        tr = T.CCATransformer(n_components).fit(x_ref, y)
    elif kind == "msn":
        tr = T.CCorATransformer(n_components).fit(x_ref, y)
    elif kind == "mahalanobis":
        tr = T.MahalanobisTransformer().fit(x_ref)
    elif kind == "euclidean":
        tr = T.StandardScalerWithDOF(ddof=1).fit(x_ref)
    else:
        raise ValueError(kind)
    center, scale, proj = tr.affine_params()
    x_ref_t = affine_transform_host(x_ref, center, scale, proj, device=device)
    eng = KNNEngine(x_ref_t, y, device=device)
    eng.set_affine(d_in, center, scale, proj)
    return eng, x_ref_t, (center, scale, proj), y, time.perf_counter() - t0


def gather_cuts(nq, n_chunks):
    """Row ranges of the calls one rank's share is split into (N > 1), cut at whole rounds of the pre-filter grid."""
    round_rows = 256 * 1024
    n_rounds = -(-nq // round_rows)
    weights = {1: (), 2: (0.6,), 3: (0.45, 0.33), 4: (0.31, 0.28, 0.23)}[n_chunks]
    cuts, acc_w = [0], 0.0
    for w_ in weights:
        acc_w += w_
        cuts.append(min(nq, max(cuts[-1], int(round(acc_w * n_rounds)) * round_rows)))
    cuts.append(nq)
    return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def timed(fn, torch, steps=2, warmup=1):
    """Mean wall time of `fn()`.  The previous call's result is released BEFORE the clock starts: returning 800 MB of
    host arrays to the OS takes ~35 ms, which belongs to the caller's `del`, not to the call being timed (the call's
    own allocation and first-touch page faults stay inside)."""
    out = None
    for _ in range(warmup):
        out = None
        out = fn()
    torch.cuda.synchronize()
    total = 0.0
    for _ in range(steps):
        out = None
        t0 = time.perf_counter()
        out = fn()
        total += time.perf_counter() - t0
    t0 = time.perf_counter()
    torch.cuda.synchronize()  # (device-resident calls are asynchronous: one wait for all of them, as before)
    total += time.perf_counter() - t0
    return total / steps, out


def oracle_slice_check(x_ref_t, affine, q_raw_slice, k, dist, idx, formula="expanded", pred=None, y=None, weights=None):
    """Rows [0, n) of a run against the oracle on the same inputs."""
    from oracle import oracle as O

    c, s, p = affine
    q_t = O.affine(q_raw_slice, c, s, p)
    od, oi = O.kneighbors(x_ref_t, q_t, k, formula)
    res = {"rows": int(len(q_raw_slice)), "index_rows_equal": int((idx == oi).all(axis=1).sum()),
           "dist_bit_equal": bool(np.array_equal(dist, od)) if dist is not None else None}
    if pred is not None:
        want = O.predict(y, od, oi, weights)
        res["max_rel_pred_err"] = float((np.abs(pred - want) / np.maximum(np.abs(want), 1e-300)).max())
    return res


def cpu_reference(x_ref_t, affine, q_raw_sample, k, algorithm="auto"):
    """The reference's CPU path on this host: numpy transform + sklearn kneighbors (what
    RawKNNRegressor.kneighbors calls, REF _base.py:162-164) + the restated reorder."""
    from oracle import oracle as O

    center, scale, proj = affine
    info = {"unit": "Mqueries/s", "sample": f"{len(q_raw_sample)} query rows of the same workload (prefix of rank 0's batch)"}
    try:
        import sklearn
        from sklearn.neighbors import KNeighborsRegressor
        from threadpoolctl import threadpool_info

        reg = KNeighborsRegressor(n_neighbors=k, algorithm=algorithm).fit(x_ref_t, np.zeros(len(x_ref_t)))
        t0 = time.perf_counter()
        q_t = q_raw_sample
        if center is not None:
            q_t = q_t - center
        if scale is not None:
            q_t = q_t / scale
        if proj is not None:
            q_t = q_t @ proj
        dist, idx = reg.kneighbors(q_t)
        dist, idx = O.deterministic_reorder(dist, idx)
        dt = time.perf_counter() - t0
        threads = [p.get("num_threads") for p in threadpool_info() if p.get("user_api") == "openmp"]
        single = reg._fit_method != "brute"  # sklearn's tree queries run on one thread unless n_jobs is set
        info.update(value=len(q_raw_sample) / dt / 1e6, cores=1 if single else int(max(threads) if threads else os.cpu_count()),
                    kind="reference", engine=f"scikit-learn {sklearn.__version__} algorithm={algorithm!r} -> {reg._fit_method} + restated reorder",
                    host_cpus=os.cpu_count(), seconds=dt)
        return info, (dist, idx)
    except ImportError as err:  # sklearn missing on the box: time the C restatement instead
        t0 = time.perf_counter()
        q_t = O.affine(q_raw_sample, center, scale, proj)
        dist, idx = O.kneighbors(x_ref_t, q_t, k, "expanded")
        dt = time.perf_counter() - t0
        info.update(value=len(q_raw_sample) / dt / 1e6, cores=O.num_threads(), kind="port",
                    engine=f"oracle/knn_oracle.c (OpenMP); sklearn unavailable: {err}", host_cpus=os.cpu_count(), seconds=dt)
        return info, (dist, idx)


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file travels with the repository)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8") as fh:
            return json.load(fh)["metric"]
    except (OSError, KeyError, ValueError):
        return "Mqueries/sec + achieved HBM GB/s, 10M\u00d750k\u00d732 k=5, 1/2/4/8 MI355X"


def library_sha16():
    """First 16 hex digits of the SHA-256 of the loaded HIP library: the committed counter summaries carry the stamp of
    the build they were measured on and are only quoted for that build."""
    import hashlib

    from sknnr_amd import _native

    try:
        with open(_native.library_path(), "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def measure_traffic(args, argv):
    """HBM bytes per step, counted IN THIS RUN: the same command (headline only, three steps) as a child process under
    `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` -- separate passes, counters only, as
    /opt/skills/guides/MI355X_MICROARCH.md (HBM, rocprofv3 PMC slots) prescribes: FETCH_SIZE takes 3 of the 4 TCC slots,
    WRITE_SIZE 2; both print KiB; gfx950 counts a wide coalesced streaming read at half its bytes, so FETCH_SIZE is doubled
    for the pre-filter (its reads are the LDS-DMA stage copies) and taken as read for the thread-per-row / gather kernels.
    Called before this process imports torch (a profiler child must not be started from a process that holds the GPU)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None, "rocprofv3 is not on PATH"
    keep = []
    skip_next = False
    for a in argv:  # the child: this run's shape arguments, headline only
        if skip_next:
            skip_next = False
            continue
        if a in ("--steps", "--warmup", "--traffic", "--gpus"):
            skip_next = True
            continue
        if a.startswith(("--steps=", "--warmup=", "--traffic=", "--gpus=")) or a in ("--no-extras", "--no-cpu-baseline"):
            continue
        keep.append(a)
    child = [sys.executable, os.path.abspath(__file__), *keep, "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
             "--traffic", "off", "--traffic-child"]
    sums, launches = {}, {}
    tmp = tempfile.mkdtemp(prefix="sknnr_traffic_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR="/tmp")
            # (a process group of its own: a pass that overruns is ended with everything it started -- nothing of it may
            #  still be on the GPU when the timed region of this process begins)
            proc = subprocess.Popen([rocprof, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--", *child],
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp", start_new_session=True)
            try:
                _, err = proc.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                import signal

                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.communicate()
                return None, f"the {counter} pass did not finish in 180 s"
            if proc.returncode != 0:
                return None, f"the {counter} pass failed (rc {proc.returncode}): {err[-300:]}"
            rows = []
            for f in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
                rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "sknnr" in r["Kernel_Name"]]
            if not rows:
                return None, f"the {counter} pass wrote no counter rows"
            grid_max = max(int(r["Grid_Size"]) for r in rows if "coarse" in r["Kernel_Name"])
            for r in rows:
                short = r["Kernel_Name"].split("(")[0].replace("void ", "")
                sums.setdefault(short, {}).setdefault(counter, 0.0)
                sums[short][counter] += float(r["Counter_Value"])
                if "coarse" in r["Kernel_Name"] and int(r["Grid_Size"]) == grid_max:
                    launches.setdefault(counter, []).append(float(r["Counter_Value"]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    n_steps = len(launches["FETCH_SIZE"])  # bulk launches of the pre-filter = steps of the child (warm-up included)
    if n_steps == 0 or len(launches["WRITE_SIZE"]) != n_steps:
        return None, "the two passes saw different launch counts"
    dominant = (2.0 * sum(launches["FETCH_SIZE"]) + sum(launches["WRITE_SIZE"])) * 1024.0 / n_steps
    step = sum((2.0 if "coarse" in kn else 1.0) * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0) for kn, c in sums.items()) * 1024.0 / n_steps
    by_kernel = {kn: ((2.0 if "coarse" in kn else 1.0) * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0 / n_steps for kn, c in sums.items()}
    return {"dominant_bytes_per_launch": dominant, "step_bytes": step, "steps_counted": n_steps,
            "by_kernel_bytes_per_step": {k: v for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])[:8]}}, (
        "counted in this run: two child passes of this command under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE; counters only), "
        "FETCH_SIZE x2 for the pre-filter's wide LDS-DMA reads (gfx950 tallies them at half), KiB -> bytes, per bulk launch / per step")


def committed_traffic(n_ref, d_t, k):
    """HBM bytes per query row from the committed rocprofv3 --pmc passes of this same command (counters cannot be
    collected inside the process): (dominant kernel, whole step, note), or (None, None, why)."""
    path = os.path.join(ROOT, "profiles", "r04_coarse_pmc.json")
    if (n_ref, d_t, k) != (50_000, 32, 5):
        return None, None, "no committed counter passes for this shape"
    if not os.path.exists(path):
        return None, None, "profiles/r04_coarse_pmc.json is missing"
    pmc = json.load(open(path))
    if pmc.get("lib_sha16") != library_sha16():
        return None, None, (f"profiles/r04_coarse_pmc.json was measured on library {pmc.get('lib_sha16')}, this run loads "
                            f"{library_sha16()}: not quoted")
    return pmc["hbm_bytes_per_query_row"], pmc.get("step_hbm_bytes_per_query_row"), (
        "profiles/r04_coarse_pmc.json (rocprofv3 --pmc, separate passes; FETCH_SIZE x2 only for the kernels whose reads are "
        "wide coalesced: the pre-filter's LDS-DMA stages; WRITE_SIZE as read), same library build, scaled to this step's rows")


def mfma_frac(nq, n_ref, d_t, coarse_ms):
    tf = 2.0 * nq * n_ref * d_t / (coarse_ms * 1e-3) / 1e12 if coarse_ms > 0 else 0.0
    return tf, tf / PEAK_F16_MFMA_TFLOPS


def extra_config(name, kind, nq, n_ref, d_in, k, torch, device, *, t=40, n_components=None, predict=None,
                 dataframe_ids=False, law="baseline", x_ref=None, check_rows=100_000):
    """One BASELINE configuration on one GPU, device-resident input: wall and kernel time, roofline
    fraction of its pre-filter, and a slice checked against the oracle."""
    eng, x_ref_t, affine, y, fit_s = fit_space(kind, n_ref, d_in, t, device, n_components, x_ref=x_ref)
    d_t = x_ref_t.shape[1]
    formula = "direct" if d_t <= 15 else "expanded"  # what algorithm="auto" resolves to (SKL/neighbors/_base.py:620-648)
    q = gen_queries(nq, d_in, 4242, torch, law)
    ids_table = torch.arange(n_ref, device="cuda", dtype=torch.int64) + 100_000 if dataframe_ids else None

    def run():
        if predict:
            return eng.predict(q, k, predict, apply_affine=True, formula=formula)
        dist, idx = eng.kneighbors(q, k, apply_affine=True, formula=formula)
        if dataframe_ids:  # REF _base.py:177-180 on the device
            idx = eng.crosswalk(idx, np.arange(n_ref, dtype=np.int64) + 100_000)
        return dist, idx

    small = nq <= 2_000_000  # a call of a few ms: more passes, and the clocks are up before the first timed one
    for _ in range(4 if small else 1):
        run()
    torch.cuda.synchronize()
    # The MEDIAN pass (by wall time) is reported, with the kernel timings of that pass: the boxes of the pool show the odd
    # pass 30-40 % slower than its neighbours (C5 25.3 / 36.6 ms on consecutive runs of one build), and a mean of two
    # passes would carry it.
    passes = []
    for _ in range(7 if small else 3):
        eng.reset_stats()
        wall, out = timed(run, torch, steps=1, warmup=0)
        passes.append((wall, eng.stats()))
    passes.sort(key=lambda p: p[0])
    wall, st = passes[len(passes) // 2]
    coarse_ms = st["total_coarse_ms"] / max(1, st["timed_calls"])
    kernel_ms = st["total_kernel_ms"] / max(1, st["timed_calls"])
    # (rows the timed pre-filter launches processed: the thin last round that runs beside the finaliser is not timed)
    tf, frac = mfma_frac(st["coarse_rows_timed"] / max(1, st["timed_calls"]), n_ref, d_t, coarse_ms)
    n_chk = min(check_rows, nq)
    q_host = q[:n_chk].cpu().numpy()
    if predict:
        dist_c, idx_c = eng.kneighbors(q[:n_chk].contiguous(), k, apply_affine=True, formula=formula)
        chk = oracle_slice_check(x_ref_t, affine, q_host, k, dist_c.cpu().numpy(), idx_c.cpu().numpy(), formula,
                                 pred=out[:n_chk].cpu().numpy(), y=y, weights=predict)
    else:
        idx_h = out[1][:n_chk].cpu().numpy()
        if dataframe_ids:
            idx_h = idx_h - 100_000
        chk = oracle_slice_check(x_ref_t, affine, q_host, k, out[0][:n_chk].cpu().numpy(), idx_h, formula)
    res = {"workload": name, "Mq_s": nq / wall / 1e6, "ms": wall * 1e3, "kernel_ms": kernel_ms, "prefilter_ms": coarse_ms,
           "prefilter_TFLOPs": tf, "frac": frac, "d_t": int(d_t), "formula": formula,
           "exact_fallbacks_per_pass": int(st["exact_fallbacks"] / max(1, st["timed_calls"])),
           "fit_seconds": fit_s, "passes_ms": [p[0] * 1e3 for p in passes], "oracle_check": chk}
    assert frac <= 1.0, res
    eng.close()
    del q
    torch.cuda.empty_cache()
    return res


def hamming_config(torch, nq=200_000, n_ref=20_000, n_trees=500, k=5, levels=300, check_rows=1500):
    """RFNN-shaped workload (REF src/sknnr/_weighted_trees.py:53-59, :139-140): weighted Hamming distance over the node
    ids of `n_trees` trees, real-valued tree weights, device-resident ids.  Queries are perturbed copies of reference
    rows (trees agree far more often than random ids do).  Bound: VALU issue -- the integer pre-filter spends three
    vector instructions per two (pair, tree) compares (hamming.hip.h)."""
    from oracle import oracle as O
    from sknnr_amd import _native as N

    rng = np.random.default_rng(0)
    ref = rng.integers(0, levels, (n_ref, n_trees)).astype(np.float64)
    src = rng.integers(0, n_ref, nq)
    q = np.where(rng.random((nq, n_trees)) < 0.6, ref[src], rng.integers(0, levels, (nq, n_trees)).astype(np.float64))
    w = rng.random(n_trees) + 0.01
    ix = N.Index(ref)
    ix.set_hamming_weights(w)
    o = ix.make_opts(k, formula=N.FORMULA_HAMMING)
    qd = torch.as_tensor(q, device="cuda")
    dd = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    di = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    passes = []
    for rep in range(4):  # the first pass warms up; the median of the other three is reported
        ix.reset_stats()
        t0 = time.perf_counter()
        ix.kneighbors_device(qd.data_ptr(), nq, o, dd.data_ptr(), di.data_ptr())
        torch.cuda.synchronize()
        if rep:
            passes.append((time.perf_counter() - t0, ix.stats()))
    passes.sort(key=lambda p: p[0])
    wall, st = passes[1]
    n_chk = min(check_rows, nq)
    od, oi = O.kneighbors_hamming(ref, q[:n_chk], w, k)
    compares = float(n_ref) * nq * n_trees
    # 256 CUs x 4 SIMDs x 64 lanes, one wave instruction per 4 clocks at 2.4 GHz, 1.5 instructions per compare
    valu_peak = 256 * 4 * 64 * 2.4e9 / 4 / 1.5
    res = {"workload": f"RFNN-shaped weighted Hamming: {nq} queries x {n_ref} refs x {n_trees} trees, k={k} (integer pre-filter + float64 re-score)",
           "Mq_s": nq / wall / 1e6, "ms": wall * 1e3, "kernel_ms": st["total_kernel_ms"],
           "compares_per_s": compares / (st["total_kernel_ms"] * 1e-3), "bound": "valu",
           "valu_peak_compares_per_s": valu_peak, "frac": compares / (st["total_kernel_ms"] * 1e-3) / valu_peak,
           "algorithmic_bytes": float(nq) * n_trees * 8 + float(n_ref) * n_trees * 8 + nq * k * 16,
           # `frac` prices the kernel against ITS OWN instruction mix (it says the loop is tight, not that the problem is near a
           # hardware bound); the problem's own roofs: moving its bytes once at 8 TB/s, and one compare per lane and clock
           "hbm_bound_ms": (float(nq) * n_trees * 8 + float(n_ref) * n_trees * 8 + nq * k * 16) / 8e12 * 1e3,
           "frac_of_hbm_bound": (float(nq) * n_trees * 8 + float(n_ref) * n_trees * 8 + nq * k * 16) / 8e12 / (st["total_kernel_ms"] * 1e-3),
           "frac_of_one_compare_per_lane_clock": compares / (st["total_kernel_ms"] * 1e-3) / (256 * 4 * 16 * 2.4e9),
           "frac_note": "frac = share of the vector-issue rate at this kernel's 1.5 instructions per compare; the algorithmic-bytes and "
                        "one-compare-per-lane-clock fractions beside it are the problem's own roofs (VERDICT r3, weak 8)",
           "oracle_check": {"rows": n_chk, "index_rows_equal": int((di[:n_chk].cpu().numpy() == oi).all(axis=1).sum()),
                            "dist_bit_equal": bool(np.array_equal(dd[:n_chk].cpu().numpy(), od))}}
    ix.close()
    del qd, dd, di
    torch.cuda.empty_cache()
    return res


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))
    # HBM traffic counted in-run (profiler children first: this process has not touched the GPU yet)
    measured, measured_note = None, None
    world_env = int(os.environ.get("WORLD_SIZE", 1))
    if args.traffic in ("auto", "measure") and world_env == 1 and not args.force_dist and not args.traffic_child:
        import shutil

        under_profiler = any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if under_profiler and args.traffic == "auto":
            measured_note = "this run is itself under a profiler: no nested counter passes"
        elif args.traffic == "measure" or shutil.which("rocprofv3"):
            measured, measured_note = measure_traffic(args, argv)
    # Libraries print to stdout (RCCL's version banner at communicator creation, for one): the contract is ONE JSON line
    # there, so file descriptor 1 points at stderr until that line is written.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(line, flush=True)

    import torch

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        args.gpus = world
    device = local_rank % max(1, torch.cuda.device_count())  # (--backend gloo: the ranks may share a GPU)
    torch.cuda.set_device(device)
    import torch.distributed as dist

    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from sknnr_amd.distributed import cyclic_slot

    # ---- fit (host, once): CCA ordination of the synthetic reference set -> affine map ------
    eng, x_ref_t, affine, y_ref, t_fit = fit_space("gnn", args.refs, args.dims, args.targets, device)
    center, _, proj = affine
    d_t = proj.shape[1]

    # strong scaling: the job's rows are split over the ranks; weak: every rank gets --rows rows
    nq = args.rows // world if args.scaling == "strong" else args.rows
    total_rows = nq * world
    k = args.k
    q = gen_queries(nq, args.dims, 1000 + rank, torch)

    # N > 1: two all-gather chunks (the first travels while the second is computed), cut at a whole round of the
    # pre-filter grid (256 CUs x 1024 rows per workgroup).  Every chunk is a call of its own: it drains the device at its
    # end and pays the fixed part of the finaliser / exact scan again -- measured on one GPU with one rank's share of the job
    # (scripts/rank_share_probe.sh, profiles/r03_rank_share.txt): four chunks cost +3.3 / +1.2 / +1.8 ms at N = 2 / 4 / 8
    # against 25.6 / 14.0 / 6.8 ms for the share as one call, about what the gather they hide takes (800 MB x (N-1)/N per
    # rank over xGMI).  Two chunks halve that and leave 40 % of the gather in the open.  (--gather-chunks)
    gather_chunks = gather_cuts(nq, args.gather_chunks)
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
                    # in place (the send buffer is this rank's slot of the receive buffer) when the probe below said
                    # the backend takes it
                    works.append(dist.all_gather_into_tensor(whole, own if gather_in_place[0] else own.clone(), async_op=True))
        for w in works:
            w.wait()
        torch.cuda.current_stream().wait_stream(comm_stream)
        return d_all, i_all

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist and not args.no_gather:
        # One probe collective BEFORE the timed region decides whether the all-gather may be in place: a synchronous
        # refusal and an asynchronous failure (surfacing at wait / synchronize) are both caught here, and every rank
        # takes the same decision (MIN over ranks).
        probe = torch.arange(world * 8, dtype=torch.float64, device="cuda").reshape(world * 2, 4)
        want = probe.clone()
        ok = 1
        try:
            w_ = dist.all_gather_into_tensor(probe, probe[rank * 2: rank * 2 + 2], async_op=True)
            w_.wait()
            torch.cuda.synchronize()
            ok = int(torch.equal(probe, want))
        except RuntimeError:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_in_place[0] = bool(flag.item())
    for _ in range(args.warmup):
        step()
    barrier()
    eng.reset_stats()  # kernel timings below are summed over every call of the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d_out, i_out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    st = eng.stats()
    coarse_ms = st["total_coarse_ms"] / args.steps   # pre-filter kernel, per step (all chunks of the step)
    kernel_ms = st["total_kernel_ms"] / args.steps

    t_max = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rows * args.steps / elapsed / 1e6

    if rank == 0:
        # this rank's rows through this rank's kernel: the rows of the launches that coarse_ms sums (when the thin last
        # round of a call runs beside the finaliser, only the 16-wave bulk launch is timed, and only its rows are priced)
        rows_timed = st["coarse_rows_timed"] / args.steps
        achieved_tf, frac = mfma_frac(rows_timed, args.refs, d_t, coarse_ms)
        alg_bytes = nq * args.dims * 8 + args.refs * d_t * 8 + nq * k * 16
        if measured is not None:
            # per launch of the dominant kernel / per step, as counted (the child ran the same rows per step)
            traffic, traffic_step, traffic_note, traffic_source = measured["dominant_bytes_per_launch"], measured["step_bytes"], measured_note, "measured"
        elif args.traffic == "off":
            traffic, traffic_step, traffic_note, traffic_source = None, None, "not requested (--traffic off)", None
        else:
            per_row, step_per_row, traffic_note = committed_traffic(args.refs, d_t, k)
            traffic = per_row * rows_timed if per_row is not None else None
            traffic_step = step_per_row * nq if step_per_row is not None else None
            traffic_source = "committed" if traffic is not None else None
            if measured_note:
                traffic_note = f"in-run count unavailable ({measured_note}); " + traffic_note
        assert frac <= 1.0, (frac, coarse_ms, st)
        roofline = {
            "kernel": "sknnr::coarse2_kernel<KS=%d,M=%d,WAVES=16> (f16 split MFMA pre-filter: seeded thresholds, main hi.hi products swept on the matrix "
                      "pipe with the skip test in their shadow, hits corrected (lo.hi + hi.lo) in the batched flush; lane-local top-M)" % ((d_t + 15) // 16, 6 if k <= 5 else 8),
            "bound": "mfma", "achieved": achieved_tf, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": frac, "traffic": traffic, "traffic_source": traffic_source, "traffic_note": traffic_note,
            "traffic_by_kernel_bytes_per_step": measured["by_kernel_bytes_per_step"] if measured else None,
            "traffic_all_kernels_of_the_step": traffic_step,
            "traffic_over_algorithmic_bytes_step": (traffic_step / alg_bytes) if traffic_step else None,
            "algorithmic_bytes_step": alg_bytes,
            "executed_over_algorithmic_mfma": st.get("mfma_executed_ratio"),
            "vs_f32_mfma_peak": achieved_tf / PEAK_F32_MFMA_TFLOPS,
            "kernel_ms_per_step": coarse_ms, "kernel_rows_per_step": rows_timed, "all_kernels_ms_per_step": kernel_ms,
            "timed_calls": int(st["timed_calls"]),
            "hbm_algorithmic_GBs": alg_bytes / (kernel_ms * 1e-3) / 1e9,
            "hbm_frac_of_peak": alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
        }
        result = {
            "metric": baseline_metric(),
            "value": value, "unit": "Mqueries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f16 split MFMA (hi.hi swept, lo.hi + hi.lo corrections; f32 accumulate) pre-filter + f64 exact re-score",
            "data": "synthetic",
            "config": {
                "workload": f"GNN-style kneighbors: affine {args.dims}->{d_t} (CCA fit on synthetic refs) + "
                            f"{total_rows} query rows in total ({nq} per GPU, {args.scaling} scaling) x {args.refs} refs x "
                            f"{d_t} dims, k={k}, deterministic reorder, float64 (dist, idx) out"
                            + ((" + RCCL all-gather" if args.backend == "nccl" else " + gloo all-gather (rehearsal: ranks share a GPU)")
                               if use_dist and not args.no_gather else ""),
                "total_rows": total_rows, "rows_per_gpu": nq, "n_ref": args.refs, "d_in": args.dims, "d_t": int(d_t), "k": k,
                "parallelism": f"query-row shards x{world}",
            },
            "roofline": roofline,
            "library_sha16": library_sha16(),
            "fit_seconds": t_fit,
            "exact_fallbacks_per_step": int(st["exact_fallbacks"] / args.steps), "queries_answered": int(st["queries"]),
        }
        if use_dist:
            result["gather_in_place"] = gather_in_place[0]
        if use_dist and not args.no_gather:
            # The gathered arrays hold what the OTHER ranks computed: rank 0 regenerates the last rank's rows (seeded) and
            # answers a prefix of its first gather chunk itself, at that block's global position -- equal bit for bit when
            # every block sits where the chunk-cyclic layout says and every rank ran the same index.
            r_chk = world - 1
            q_r = q if r_chk == rank else gen_queries(nq, args.dims, 1000 + r_chk, torch)
            a0, b0 = gather_chunks[0]
            n_chk = min(20_000, b0 - a0)
            lo = cyclic_slot(world, r_chk, a0, b0)
            d_exp, i_exp = eng.kneighbors(q_r[a0:a0 + n_chk], k, apply_affine=True, deterministic=True, row_offset=lo)
            result["gather_check"] = {"rank": r_chk, "rows": int(n_chk), "global_row": int(lo),
                                      "equal": bool(torch.equal(i_out[lo:lo + n_chk], i_exp) and torch.equal(d_out[lo:lo + n_chk], d_exp))}
            del q_r
        if world == 1 and not args.no_cpu_baseline:
            n_s = min(args.cpu_sample, nq)
            q_host = q[:n_s].cpu().numpy()
            base, (cd, ci) = cpu_reference(x_ref_t, affine, q_host, k)
            result["cpu_baseline"] = base
            gi = i_out[rank * nq: rank * nq + n_s].cpu().numpy() if use_dist and not args.no_gather else i_out[:n_s].cpu().numpy()
            gd = d_out[rank * nq: rank * nq + n_s].cpu().numpy() if use_dist and not args.no_gather else d_out[:n_s].cpu().numpy()
            rel = np.abs(gd - cd) / np.maximum(np.abs(cd), 1e-300)
            result["parity_vs_cpu_reference"] = {
                "rows": int(n_s), "index_rows_equal": int((gi == ci).all(axis=1).sum()),
                "max_rel_dist_err": float(rel.max()),
            }
            result["speedup_vs_cpu"] = value / base["value"]
        if world == 1 and not args.no_extras and not use_dist:
            del d_out, i_out
            result.update(extras(args, eng, q, x_ref_t, affine, torch, device))
        emit(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


def extras(args, eng, q, x_ref_t, affine, torch, device):
    """Everything BASELINE.json / SURVEY.md 8(d) ask for beyond the headline line, at one GPU."""
    out = {}
    nq, k = q.shape[0], args.k

    # ---- what to expect of the strong-scaling runs (the driver's SCALE file): one rank's share of the job, measured here --
    # (query-row sharding: no data-path collective before the final all-gather; N = 8 moves 7 x 100 MB per rank over its
    #  seven xGMI links, ~153 GB/s each: < 1 ms, issued chunk-wise under the kernels)
    pred = {}
    for n_gpu in (2, 4, 8):
        share = nq // n_gpu
        wall, _ = timed(lambda: eng.kneighbors(q[:share], k, apply_affine=True, deterministic=True), torch, steps=3, warmup=1)
        chunks = gather_cuts(share, args.gather_chunks)
        wall_c, _ = timed(lambda: [eng.kneighbors(q[a:b], k, apply_affine=True, deterministic=True, row_offset=a) for a, b in chunks][-1],
                          torch, steps=3, warmup=1)
        pred[str(n_gpu)] = {"rows_per_gpu": share, "ms_per_share": wall * 1e3, "predicted_Mq_s_before_gather": nq / wall / 1e6,
                            "ms_per_share_in_gather_chunks": wall_c * 1e3, "gather_chunks": len(chunks),
                            "predicted_Mq_s_chunked_before_last_gather": nq / wall_c / 1e6,
                            "gather_bytes_received_per_rank": (n_gpu - 1) * share * k * 16,
                            "last_gather_bytes_received_per_rank": (n_gpu - 1) * (chunks[-1][1] - chunks[-1][0]) * k * 16}
    out["strong_scaling_expectation"] = pred

    # ---- the headline workload entered from host arrays (PCIe-inclusive; never `value`) ----------
    q_host = q.cpu().numpy()
    median = lambda xs: float(np.median(np.asarray(xs, dtype=np.float64)))  # noqa: E731
    walls = []
    for _ in range(6):  # (the first call allocates the pinned staging buffers: not counted)
        wall, (hd, hi) = timed(lambda: eng.kneighbors(q_host, k, apply_affine=True), torch, steps=1, warmup=0)
        walls.append(wall)
    out["host_to_host_Mq_s"] = nq / median(walls[1:]) / 1e6
    out["host_to_host_calls_ms"] = [w * 1e3 for w in walls]
    out["host_to_host_note"] = ("numpy rows in, fresh numpy (dist, idx) out; MEDIAN of the five calls after the first, every call's time listed; "
                                "allocation and first-touch faults of the 800 MB of outputs are inside (they vary by tens of ms with the host's page state)")
    # the same rows as float32 (what a raster usually holds): widened by the prep kernel, half the PCIe bytes in
    q_f32 = q_host.astype(np.float32)
    q_back = torch.as_tensor(q_f32.astype(np.float64), device="cuda")
    ref_d, ref_i = eng.kneighbors(q_back, k, apply_affine=True)
    walls32 = []
    for _ in range(6):
        wall, (hd32, hi32) = timed(lambda: eng.kneighbors(q_f32, k, apply_affine=True), torch, steps=1, warmup=0)
        walls32.append(wall)
    out["host_to_host_f32_Mq_s"] = nq / median(walls32[1:]) / 1e6
    out["host_to_host_f32_calls_ms"] = [w * 1e3 for w in walls32]
    out["host_to_host_f32_note"] = ("float32 numpy rows in (sknnr_query_opts.query_dtype = F32: widened on the device, exactly); median of five; "
                                    f"equal to the float64 call on the widened rows: {bool(np.array_equal(hi32, ref_i.cpu().numpy()) and np.array_equal(hd32, ref_d.cpu().numpy()))}")
    del q_f32, q_back, ref_d, ref_i, hd32, hi32
    tile = 1_000_000
    d_st = np.empty((nq, k))
    i_st = np.empty((nq, k), dtype=np.int64)

    def stream_run():
        with eng.open_stream(k, apply_affine=True) as s:
            for a in range(0, nq, tile):
                s.push(q_host[a:a + tile], out_idx=i_st[a:a + tile], out_dist=d_st[a:a + tile])
        return None

    timed(stream_run, torch, steps=1, warmup=0)  # (the warm-up pass touches the caller-owned output arrays)
    st_walls = [timed(stream_run, torch, steps=1, warmup=0)[0] for _ in range(5)]
    out["stream_Mq_s"] = nq / median(st_walls) / 1e6
    out["stream_calls_ms"] = [w * 1e3 for w in st_walls]
    out["stream_note"] = (f"{-(-nq // tile)} pushes of {tile} rows through sknnr_stream_*; equals the one-call result: "
                          f"{bool(np.array_equal(i_st, hi) and np.array_equal(d_st, hd))}")
    del d_st, i_st, hd, hi

    # ---- the same workload through the estimator surface (GNNRegressor.kneighbors(X_numpy)) -----
    import sknnr_amd
    from sknnr_amd import synth

    x_ref = synth.make_features(args.refs, args.dims, seed=0)
    y = synth.make_targets(x_ref, t=args.targets, kind="positive")
    t0 = time.perf_counter()
    est = sknnr_amd.GNNRegressor(n_neighbors=k).fit(x_ref, y)
    est_fit = time.perf_counter() - t0
    timed(lambda: est.kneighbors(q_host), torch, steps=1, warmup=0)
    est_walls = [timed(lambda: est.kneighbors(q_host), torch, steps=1, warmup=0)[0] for _ in range(5)]
    out["estimator_Mq_s"] = nq / median(est_walls) / 1e6
    out["estimator_calls_ms"] = [w * 1e3 for w in est_walls]
    out["estimator_note"] = (f"GNNRegressor(n_neighbors={k}).fit in {est_fit:.2f} s (incl. independent prediction), "
                             "kneighbors(X_numpy) wall: sklearn validate_data (no host finiteness pass) + host pipeline")
    del est, q_host
    torch.cuda.empty_cache()

    # ---- BASELINE.json configs 2-5 on one GPU --------------------------------------------------------
    cfgs = []
    cfgs.append(extra_config("C2 EuclideanKNNRegressor 1M x 10k x 16, k=5", "euclidean", 1_000_000, 10_000, 16, 5, torch, device))
    cfgs.append(extra_config("C3 GNNRegressor 10M x 50k x 32, k=7, predict weights='distance', T=40", "gnn",
                             10_000_000, 50_000, 32, 7, torch, device, predict="distance"))
    cfgs.append(extra_config("C4 MahalanobisKNNRegressor 10M x 50k x 64, k=5 (whole job on one GPU)", "mahalanobis",
                             10_000_000, 50_000, 64, 5, torch, device))
    cfgs.append(extra_config("C5 MSNRegressor(n_components=8) 6.25M (one GPU's share of 50M) x 100k x 32->8, k=1, dataframe ids",
                             "msn", 6_250_000, 100_000, 32, 1, torch, device, n_components=8, dataframe_ids=True))
    out["configs"] = cfgs
    try:
        out["hamming"] = hamming_config(torch)
    except Exception as err:  # (this line must not cost the headline)
        out["hamming"] = {"error": repr(err)}

    # ---- C5's CPU baseline: sklearn picks a single-threaded kd_tree for D_t <= 15; report brute too ----
    try:
        eng5, x5_t, aff5, _, _ = fit_space("msn", 100_000, 32, args.targets, device, 8)
        q5 = gen_queries(100_000, 32, 4242, torch).cpu().numpy()
        c5 = {}
        for alg in ("auto", "brute"):
            info, _ = cpu_reference(x5_t, aff5, q5, 1, algorithm=alg)
            c5[alg] = info
        out["cpu_baseline_c5"] = c5
        eng5.close()
    except Exception as err:  # a missing sklearn must not cost the whole line
        out["cpu_baseline_c5"] = {"error": repr(err)}

    # ---- laws hostile to the tile-level skip test (data-dependence of `value`) ---------------------
    laws = []
    rng = np.random.default_rng(0)
    cube = rng.random((args.refs, args.dims))
    laws.append(extra_config("uniform hypercube refs and queries, 10M x 50k x 32, k=5 (Euclidean space)", "euclidean",
                             10_000_000, args.refs, args.dims, 5, torch, device, law="uniform", x_ref=cube))
    srt = synth.make_features(args.refs, args.dims, seed=0)
    srt = np.ascontiguousarray(srt[np.argsort(srt[:, 0])])
    laws.append(extra_config("baseline law, reference rows sorted by feature 0, 10M x 50k x 32, k=5 (GNN space)", "gnn",
                             10_000_000, args.refs, args.dims, 5, torch, device, x_ref=srt))
    out["laws"] = laws
    return out


if __name__ == "__main__":
    main()
