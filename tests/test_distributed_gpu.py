"""The N > 1 path on real devices: two RCCL ranks (one process per GPU) run ShardedKNN over the HIP engine
-- contiguous shards with uneven blocks, the chunk-cyclic in-place all-gather of bench.py, and the
X=None self query -- and bench.py itself under torch.distributed.run.  Skipped when fewer than two
MI355X are visible (the 1-GPU box): the driver's multi-GPU run must not be the first >1-rank RCCL run."""

from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _n_devices():
    import torch

    return torch.cuda.device_count()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import sknnr_amd
        from sknnr_amd import synth
        from sknnr_amd.distributed import ShardedKNN, cyclic_slot

        x_ref, y, x_q = synth.make_problem(4000, 30001, 16, t=4, kind="positive", n_dup_queries=16)
        x_q[1] = x_q[0]
        est = sknnr_amd.GNNRegressor(n_neighbors=4, weights="distance").fit(x_ref, y)
        sh = ShardedKNN(est)
        xq_dev = torch.as_tensor(x_q, device="cuda")
        d_all, i_all = sh.kneighbors(xq_dev, 4)                      # uneven contiguous shards
        d_self, i_self = sh.kneighbors(None, 4)                      # sharded X=None
        p_all = sh.predict(xq_dev)
        n_loc, chunk = 12000, 5000                                    # chunk-cyclic, ragged last chunk
        glob = xq_dev[: world * n_loc]
        mine = torch.cat([glob[cyclic_slot(world, rank, a, min(n_loc, a + chunk)):
                               cyclic_slot(world, rank, a, min(n_loc, a + chunk)) + min(n_loc, a + chunk) - a]
                          for a in range(0, n_loc, chunk)])
        d_cyc, i_cyc = sh.kneighbors_cyclic(mine, 4, chunk_rows=chunk)
        a, b = sh.local_bounds(len(x_q))
        d_pipe, i_pipe = sh.kneighbors_pipelined(xq_dev[a:b], len(x_q), 4, chunk_rows=7000)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), d_all=d_all.cpu().numpy(), i_all=i_all.cpu().numpy(),
                 d_self=np.asarray(d_self), i_self=np.asarray(i_self), p_all=p_all.cpu().numpy(),
                 d_cyc=d_cyc.cpu().numpy(), i_cyc=i_cyc.cpu().numpy(), d_pipe=d_pipe.cpu().numpy(),
                 i_pipe=i_pipe.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_two_rccl_ranks_match_the_single_call(tmp_path):
    if _n_devices() < 2:
        pytest.skip("needs two MI355X (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp

    import sknnr_amd
    from sknnr_amd import synth

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    x_ref, y, x_q = synth.make_problem(4000, 30001, 16, t=4, kind="positive", n_dup_queries=16)
    x_q[1] = x_q[0]
    est = sknnr_amd.GNNRegressor(n_neighbors=4, weights="distance").fit(x_ref, y)
    d, i = est.kneighbors(x_q)
    ds, is_ = est.kneighbors()
    p = est.predict(x_q)
    dg, ig = est.kneighbors(x_q[: world * 12000])
    for rank in range(world):
        r = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        for got, want in ((r["i_all"], i), (r["d_all"], d), (r["i_self"], is_), (r["d_self"], ds), (r["p_all"], p),
                          (r["i_cyc"], ig), (r["d_cyc"], dg), (r["i_pipe"], i), (r["d_pipe"], d)):
            np.testing.assert_array_equal(got, want)


def test_bench_under_torch_distributed_run(tmp_path):
    """bench.py --gpus 2 exactly as the driver launches it (small job): one JSON line, strong scaling by
    default, roofline fraction <= 1 with the per-step kernel time summed over the gather chunks."""
    if _n_devices() < 2:
        pytest.skip("needs two MI355X")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # the driver's own form: bench.py starts its two ranks itself (torch.distributed.run as a child process)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "2000000"]
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["config"]["total_rows"] == 2_000_000
    assert 0 < res["roofline"]["frac"] <= 1.0 and res["roofline"]["timed_calls"] >= 2


def test_bench_force_dist_single_rank():
    """The N > 1 code path of bench.py (RCCL init, chunk-cyclic in-place all-gather on a side stream, summed
    kernel timings) with one rank -- runs on the 1-GPU box."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "2", "--warmup", "1", "--rows", "3000000",
           "--cpu-sample", "20000", "--no-extras", "--gather-chunks", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["gather_in_place"] is True and res["gather_check"]["equal"] is True
    assert res["roofline"]["timed_calls"] == 2 * 3  # three gather chunks per step (--gather-chunks; default two), all summed
    assert 0 < res["roofline"]["frac"] <= 1.0
    assert res["parity_vs_cpu_reference"]["index_rows_equal"] == res["parity_vs_cpu_reference"]["rows"]


def _gloo_worker(rank, world, port, out_dir):
    """Two (three) ranks on ONE GPU: every rank's engine lives on cuda:0, the collectives run over gloo on host tensors.
    RCCL refuses several ranks per device, so this is how the N > 1 ENGINE paths (not the RCCL transport) run on the
    one-GPU box: query-row shards, the X=None shards, reference-row shards with the device merge, the Hamming metric."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sknnr_amd
        from sknnr_amd import synth
        from sknnr_amd.distributed import RefShardedKNN, ShardedKNN

        x_ref, y, x_q = synth.make_problem(6000, 9001, 12, t=3, kind="positive", n_dup_queries=16)
        x_ref[3100] = x_ref[100]  # duplicates on both sides of the reference-shard boundaries: ties across shards
        x_ref[5100] = x_ref[100]
        x_q[1] = x_q[0]
        est = sknnr_amd.GNNRegressor(n_neighbors=4, weights="distance").fit(x_ref, y)
        sh = ShardedKNN(est)
        d_all, i_all = sh.kneighbors(x_q, 4)
        d_self, i_self = sh.kneighbors(None, 4)
        p_all = sh.predict(x_q)
        rs = RefShardedKNN(est)
        d_ref, i_ref = rs.kneighbors(x_q[:3000], 4)
        d_rself, i_rself = rs.kneighbors(None, 4)
        # the tree-node family: node ids from the host transformer, weighted Hamming on the device, both shardings
        ids_ref, ids_q = synth.make_forest_ids(3000, 500, 40, seed=5)
        w = np.random.default_rng(1).random(40) + 0.05
        ham = sknnr_amd.RawKNNRegressor(n_neighbors=3, algorithm="brute", metric="hamming", metric_params={"w": w}).fit(ids_ref, ids_ref[:, :2])
        hd, hi = ShardedKNN(ham).kneighbors(ids_q, 3)
        hrd, hri = RefShardedKNN(ham).kneighbors(ids_q, 3)
        hsd, hsi = RefShardedKNN(ham).kneighbors(None, 3)
        try:
            rs.kneighbors(x_q[:10], 6000 // world + 1)
            refused = False
        except ValueError:
            refused = True
        np.savez(os.path.join(out_dir, f"gloo_rank{rank}.npz"), d_all=d_all, i_all=i_all, d_self=d_self, i_self=i_self, p_all=p_all,
                 d_ref=d_ref, i_ref=i_ref, d_rself=d_rself, i_rself=i_rself, hd=hd, hi=hi, hrd=hrd, hri=hri, hsd=hsd, hsi=hsi,
                 refused=np.asarray(refused))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_on_one_gpu_drive_the_engine_paths(tmp_path, world):
    """the multi-rank estimator / engine paths of ShardedKNN and RefShardedKNN -- Hamming and X=None
    included -- on hardware, results equal to the single call on every rank."""
    import torch.multiprocessing as mp

    import sknnr_amd
    from sknnr_amd import synth

    mp.spawn(_gloo_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    x_ref, y, x_q = synth.make_problem(6000, 9001, 12, t=3, kind="positive", n_dup_queries=16)
    x_ref[3100] = x_ref[100]
    x_ref[5100] = x_ref[100]
    x_q[1] = x_q[0]
    est = sknnr_amd.GNNRegressor(n_neighbors=4, weights="distance").fit(x_ref, y)
    d, i = est.kneighbors(x_q)
    ds, is_ = est.kneighbors()
    p = est.predict(x_q)
    ids_ref, ids_q = synth.make_forest_ids(3000, 500, 40, seed=5)
    w = np.random.default_rng(1).random(40) + 0.05
    ham = sknnr_amd.RawKNNRegressor(n_neighbors=3, algorithm="brute", metric="hamming", metric_params={"w": w}).fit(ids_ref, ids_ref[:, :2])
    hd, hi = ham.kneighbors(ids_q)
    hsd, hsi = ham.kneighbors()
    for rank in range(world):
        r = np.load(os.path.join(str(tmp_path), f"gloo_rank{rank}.npz"))
        assert bool(r["refused"])
        for got, want in ((r["i_all"], i), (r["d_all"], d), (r["i_self"], is_), (r["d_self"], ds), (r["p_all"], p),
                          (r["i_ref"], i[:3000]), (r["d_ref"], d[:3000]), (r["i_rself"], is_), (r["d_rself"], ds),
                          (r["hi"], hi), (r["hd"], hd), (r["hri"], hi), (r["hrd"], hd), (r["hsi"], hsi), (r["hsd"], hsd)):
            np.testing.assert_array_equal(got, want)


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """`python bench.py --gpus 2` as the driver types it -- the bench starts its own two ranks -- with `--backend gloo`, which
    lets both ranks share the ONE GPU of the box (RCCL refuses that): the strong-scaling split, the chunk-cyclic in-place
    all-gather on a side stream, the max-over-ranks timing and the parity check of the gathered result run with W = 2 on
    hardware; only the transport differs from the measured configuration."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--rows", "2000000", "--gather-chunks", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["config"]["total_rows"] == 2_000_000
    assert res["config"]["rows_per_gpu"] == 1_000_000 and "gloo" in res["config"]["workload"]
    assert res["roofline"]["timed_calls"] == 2 * 3 and 0 < res["roofline"]["frac"] <= 1.0
    assert res["gather_check"]["rank"] == 1 and res["gather_check"]["rows"] > 0 and res["gather_check"]["equal"] is True
