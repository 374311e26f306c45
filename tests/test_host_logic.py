"""Host-side logic that needs no GPU: transformers against the reference's fitted matrices,
engine selection rules, parameter validation, dataset loaders, and the rule that the product
package never touches the oracle."""

from __future__ import annotations

import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


@pytest.mark.parametrize(
    ("name", "factory"),
    [
        ("euclidean", lambda T: T.StandardScalerWithDOF(ddof=1)),
        ("mahalanobis", lambda T: T.MahalanobisTransformer()),
        ("gnn_full", lambda T: T.CCATransformer()),
        ("gnn_reduced", lambda T: T.CCATransformer(3)),
        ("msn_full", lambda T: T.CCorATransformer()),
        ("msn_reduced", lambda T: T.CCorATransformer(3)),
    ],
)
def test_transformers_reproduce_the_reference_fit(name, factory, moscow):
    from sknnr_amd import transformers as T

    g = load_golden(f"moscow_{name}.npz")
    tr = factory(T).fit(moscow["X_train"], moscow["y_train"])
    for attr in ("mean_", "scale_", "env_center_", "projector_", "transform_"):
        if "tr_" + attr in g:
            np.testing.assert_allclose(getattr(tr, attr), g["tr_" + attr], rtol=1e-8, atol=1e-10)
    if "tr_scaler_mean_" in g:
        np.testing.assert_allclose(tr.scaler_.mean_, g["tr_scaler_mean_"], rtol=1e-12)
        np.testing.assert_allclose(tr.scaler_.scale_, g["tr_scaler_scale_"], rtol=1e-12)
    if "tr_n_components_" in g:
        assert tr.n_components_ == int(g["tr_n_components_"])
    np.testing.assert_allclose(tr.transform(moscow["X_train"]), g["Xt_train"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(tr.transform(moscow["X_test"]), g["Xt_test"], rtol=1e-8, atol=1e-10)
    # the affine triple handed to the device is the same map
    c, s, p = tr.affine_params()
    x = moscow["X_test"] - (0 if c is None else c)
    x = x / (1 if s is None else s)
    x = x if p is None else x @ p
    np.testing.assert_allclose(x, g["Xt_test"], rtol=1e-8, atol=1e-10)


def test_swo_and_synthetic_projectors(moscow):
    from sknnr_amd import synth
    from sknnr_amd import transformers as T
    from sknnr_amd.datasets import load_swo_ecoplot

    X, y = load_swo_ecoplot(return_X_y=True)
    g = load_golden("swo_msn_k5.npz")
    tr = T.CCorATransformer().fit(X, y)
    assert tr.n_components_ == 17
    np.testing.assert_allclose(tr.projector_, g["tr_projector_"], rtol=1e-7, atol=1e-10)
    x_ref, y_pos, _ = synth.make_problem(1500, 8, 16, t=20, kind="positive")
    g = load_golden("synth_est_gnn.npz")
    tr = T.CCATransformer().fit(x_ref, y_pos)
    np.testing.assert_allclose(tr.projector_, g["tr_projector_"], rtol=1e-7, atol=1e-10)
    x_ref, y_lin, _ = synth.make_problem(1500, 8, 16, t=20, kind="linear")
    g = load_golden("synth_est_msn.npz")
    tr = T.CCorATransformer(8).fit(x_ref, y_lin)
    np.testing.assert_allclose(tr.projector_, g["tr_projector_"], rtol=1e-7, atol=1e-10)


def test_transformer_api_behaviour(moscow):
    """REF tests/test_transformers.py:116-213."""
    from sknnr_amd import transformers as T

    X, y = moscow["X_train"], moscow["y_train"]
    cca = T.CCATransformer(n_components=5).fit(X, y)
    assert cca.get_feature_names_out().tolist() == [f"cca{i}" for i in range(5)]
    assert cca.transform(X).shape == (132, 5)
    cc = T.CCorATransformer(n_components=0).fit(X, y)
    assert cc.transform(X).shape == (132, 0)
    assert T.CCorATransformer().fit(X, y).get_feature_names_out()[0] == "ccora0"
    with pytest.raises(ValueError, match=r"n_components=500 must be between 0 and \d+"):
        T.CCATransformer(n_components=500).fit(X, y)
    with pytest.raises(ValueError, match="`y` must be a 2D array"):
        T.CCATransformer().fit(X, y[:, 0])
    bad = y.copy()
    bad[3] = 0.0
    with pytest.raises(ValueError, match="All row sums must be greater than 0"):
        T.CCATransformer().fit(X, bad)
    m = T.MahalanobisTransformer().fit(X)
    cov = np.cov(m.transform(X), rowvar=False)
    np.testing.assert_allclose(cov, np.eye(28), atol=1e-8)
    sc = T.StandardScalerWithDOF(ddof=1).fit(X)
    np.testing.assert_allclose(sc.scale_, X.std(axis=0, ddof=1))


def test_engine_selection_rule():
    """SKL/neighbors/_base.py:620-648."""
    from sknnr_amd._base import _resolve_fit_method

    assert _resolve_fit_method("auto", 132, 28, 5) == "brute"
    assert _resolve_fit_method("auto", 132, 6, 5) == "kd_tree"
    assert _resolve_fit_method("auto", 132, 15, 5) == "kd_tree"
    assert _resolve_fit_method("auto", 132, 16, 5) == "brute"
    assert _resolve_fit_method("auto", 10, 3, 5) == "brute"  # k >= n // 2
    assert _resolve_fit_method("brute", 1000, 3, 5) == "brute"
    assert _resolve_fit_method("kd_tree", 1000, 30, 5) == "kd_tree"
    with pytest.raises(ValueError):
        _resolve_fit_method("nope", 10, 3, 1)


def test_parameter_surface_matches_the_reference():
    import sknnr_amd as E

    raw = E.RawKNNRegressor().get_params()
    assert raw == {"algorithm": "auto", "leaf_size": 30, "metric": "minkowski", "metric_params": None,
                   "n_jobs": None, "n_neighbors": 5, "p": 2, "weights": "uniform"}
    assert E.RawKNNRegressor.DISTANCE_PRECISION_DECIMALS == 10
    assert set(E.GNNRegressor().get_params()) == set(raw) | {"n_components"}
    assert set(E.MSNRegressor(n_components=3).get_params()) == set(raw) | {"n_components"}
    assert set(E.EuclideanKNNRegressor().get_params()) == set(raw)
    assert E.__all__ == ["RawKNNRegressor", "EuclideanKNNRegressor", "MahalanobisKNNRegressor",
                         "MSNRegressor", "GNNRegressor", "RFNNRegressor", "GBNNRegressor"]  # = REF src/sknnr/__init__.py
    # REF _rfnn.py:154-186, _gbnn.py:158-192: keyword-only constructors, forest parameters + the three kNN ones
    rf = E.RFNNRegressor().get_params()
    assert rf["n_estimators"] == 50 and rf["min_samples_leaf"] == 5 and rf["forest_weights"] == "uniform"
    assert {"criterion_reg", "criterion_clf", "max_features_reg", "max_features_clf", "class_weight_clf",
            "n_neighbors", "weights", "n_jobs"} <= set(rf) and "metric" not in rf
    gb = E.GBNNRegressor().get_params()
    assert gb["n_estimators"] == 100 and gb["max_depth"] == 3 and gb["tree_weighting_method"] == "train_improvement"
    assert {"loss_reg", "loss_clf", "alpha_reg", "forest_weights", "n_neighbors", "weights", "n_jobs"} <= set(gb)
    with pytest.raises(TypeError):
        E.RFNNRegressor(5)  # keyword-only, as in the reference


def test_non_euclidean_metrics_raise():
    from sknnr_amd._base import _effective_metric

    assert _effective_metric("minkowski", 2, None) == "euclidean"
    assert _effective_metric("euclidean", 2, None) == "euclidean"
    assert _effective_metric("hamming", 2, {"w": [1.0]}) == "hamming"  # RFNN / GBNN (REF _weighted_trees.py:53-59)
    for bad in (("minkowski", 1, None), ("manhattan", 2, None), (lambda a, b: 0.0, 2, None),
                ("minkowski", 2, {"w": [1.0]}), ("hamming", 2, {"V": 1})):
        with pytest.raises(NotImplementedError):
            _effective_metric(*bad)


def test_datasets():
    """REF tests/test_datasets.py:21-28 shapes."""
    from sknnr_amd.datasets import load_moscow_stjoes, load_swo_ecoplot

    m = load_moscow_stjoes()
    assert m.data.shape == (165, 28) and m.target.shape == (165, 35) and m.index.dtype == np.int64
    s = load_swo_ecoplot()
    assert s.data.shape == (3005, 18) and s.target.shape == (3005, 25)
    X, y = load_swo_ecoplot(return_X_y=True, as_frame=True)
    assert list(X.index[:3]) == list(s.index[:3]) and X.shape == (3005, 18)
    assert repr(m) == "Dataset(n=165, features=28, targets=35)"


def test_product_never_touches_the_oracle_or_a_cpu_engine():
    """The oracle is test infrastructure; the product path must not import it, nor
    scikit-learn's neighbour engines."""
    pkg = os.path.join(ROOT, "sknnr_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            text = open(os.path.join(dirpath, f)).read()
            if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "knn_oracle" in text:
                offenders.append((f, "oracle"))
            if re.search(r"sklearn\.neighbors|KNeighborsRegressor\s*\(|pairwise_distances|cdist\(", text):
                if f not in ("_base.py",) or re.search(r"^\s*(from|import)\s+sklearn\.neighbors", text, flags=re.M):
                    offenders.append((f, "cpu engine"))
    assert not offenders, offenders


def test_synthetic_generators_are_deterministic():
    from sknnr_amd import synth

    a = synth.make_features(100, 8, seed=1)
    b = synth.make_features(100, 8, seed=1)
    np.testing.assert_array_equal(a, b)
    x, y, q = synth.make_problem(200, 50, 8, t=5, kind="positive", n_dup_refs=10, n_dup_queries=5)
    assert (y > 0).all() and np.array_equal(x[-10:], x[:10]) and np.array_equal(q[0], x[0])


def test_bench_gather_cuts_cover_the_share_in_whole_rounds():
    """bench.py, N > 1: the calls a rank's share is split into are contiguous, cover it exactly, and every cut but the last
    falls on a whole round of the pre-filter grid (256 workgroups x 1024 rows)."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for nq in (1, 100, 262_144, 1_250_000, 2_500_000, 5_000_000, 10_000_000):
        for n_chunks in (1, 2, 3, 4):
            cuts = bench.gather_cuts(nq, n_chunks)
            assert cuts[0][0] == 0 and cuts[-1][1] == nq and 1 <= len(cuts) <= n_chunks
            for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
                assert b == c and a < b
            for a, b in cuts[:-1]:
                assert b % (256 * 1024) == 0
    assert bench.gather_cuts(1_250_000, 2) == [(0, 786_432), (786_432, 1_250_000)]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` (N > 1) outside a launcher builds a torch.distributed.run child command with one rank
    per GPU on 127.0.0.1 and the same bench arguments; under a launcher (WORLD_SIZE set) it does not."""
    import json
    import subprocess
    import sys

    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "5", "--warmup", "2", "--dry-launch", "--master-port", "29999"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    tail = cmd[cmd.index(bench):]
    assert tail == [bench, "--gpus", "8", "--steps", "5", "--warmup", "2", "--master-port", "29999"]

    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", bench)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.launch_command(["--gpus", "2", "--dry-launch"], 2, 1234)[-2:] == ["--gpus", "2"]
    src = open(bench).read()
    launch_at, torch_at = src.index("self_launch(args, argv)"), src.index("    import torch\n")
    assert launch_at < torch_at, "the parent must start its ranks before it imports torch / touches the GPU"
    assert "os.exec" not in src


@pytest.mark.parametrize("deterministic", [True, False])
@pytest.mark.parametrize("self_query", [False, True])
def test_numpy_tie_replay_equals_scikit_learns_brute_hamming_search(deterministic, self_query):
    """hamming_tie_policy("numpy"): the host-side replay of the reference's selection (argpartition, argsort, X=None
    self removal, sknnr's reorder) on full distance rows gives what scikit-learn's brute Hamming search -- the call the
    reference makes, REF _weighted_trees.py:53-59 -> SKL/neighbors/_base.py:733-760, :936-963 -- followed by
    REF _base.py:166-175 gives, on inputs where nearly every row has exact ties (three node ids, integer weights)."""
    from scipy.spatial.distance import cdist
    from sklearn.neighbors import KNeighborsRegressor

    from sknnr_amd._base import replay_reference_selection

    rng = np.random.default_rng(5)
    ref = rng.integers(0, 3, (300, 14)).astype(np.float64)
    ref[40:48] = ref[7]  # more duplicates of a row than neighbours asked for (the dup_gr_nbrs corner of the X=None path)
    q = rng.integers(0, 3, (80, 14)).astype(np.float64)
    w = rng.integers(1, 4, 14).astype(np.float64)
    for k in (1, 5, 9):
        skl = KNeighborsRegressor(n_neighbors=k, algorithm="brute", metric="hamming", metric_params={"w": w}).fit(ref, np.zeros(300))
        d, i = skl.kneighbors(None if self_query else q)
        if deterministic:  # REF src/sknnr/_base.py:166-175
            rounded = np.round(d / np.maximum(d.max(axis=1, keepdims=True), 1.0), decimals=10)
            order = np.lexsort((i, np.abs(i - np.arange(len(i))[:, None]), rounded), axis=1)
            d, i = np.take_along_axis(d, order, axis=1), np.take_along_axis(i, order, axis=1)
        full = cdist(ref if self_query else q, ref, "hamming", w=w)
        gd, gi = replay_reference_selection(full, k + self_query, np.arange(len(full)), self_query, deterministic, 10)
        np.testing.assert_array_equal(gi, i)
        np.testing.assert_array_equal(gd, d)


def test_hamming_tie_policy_setting():
    import sknnr_amd

    assert sknnr_amd.get_hamming_tie_policy() == "lowest_index"
    with sknnr_amd.hamming_tie_policy("numpy"):
        assert sknnr_amd.get_hamming_tie_policy() == "numpy"
    assert sknnr_amd.get_hamming_tie_policy() == "lowest_index"
    with pytest.raises(ValueError, match="must be one of"):
        sknnr_amd.set_hamming_tie_policy("scalar")


def test_bench_counts_hbm_traffic_from_counter_passes(tmp_path, monkeypatch):
    """bench.py --traffic: the two `rocprofv3 --pmc` child passes (FETCH_SIZE, WRITE_SIZE -- counters only, no trace domain on
    the command line) and their reduction: KiB -> bytes, FETCH_SIZE doubled for the pre-filter only, per bulk launch of the
    dominant kernel and per step.  A stub `rocprofv3` on PATH writes the CSVs a real pass would."""
    import importlib.util
    import stat
    import sys

    stub = tmp_path / "rocprofv3"
    stub.write_text(
        "#!/bin/bash\n"
        'echo "$@" >> "%s/calls.txt"\n' % tmp_path +
        'while [ $# -gt 0 ]; do case "$1" in --pmc) c=$2; shift 2;; -d) d=$2; shift 2;; --) break;; *) shift;; esac; done\n'
        'mkdir -p "$d/host/1"\n'
        'f="$d/host/1/9_counter_collection.csv"\n'
        'echo "Dispatch_Id,Kernel_Name,Grid_Size,Counter_Name,Counter_Value" > "$f"\n'
        'for i in 1 2 3 4; do\n'
        '  if [ "$c" = FETCH_SIZE ]; then v=1000; w=300; else v=100; w=700; fi\n'
        '  echo "$i,\\"void sknnr::coarse2_kernel<2, 6, 16, 0>(char const*)\\",9961472,$c,$v" >> "$f"\n'
        '  echo "$i,\\"void sknnr::coarse2_kernel<2, 6, 4, 0>(char const*)\\",40960,$c,10" >> "$f"\n'
        '  echo "$i,sknnr::prep_queries_direct_kernel<2>(sknnr::PrepArgs),10000128,$c,$w" >> "$f"\n'
        '  echo "$i,at::native::some_torch_kernel,64,$c,99999" >> "$f"\n'
        "done\n")
    stub.chmod(stub.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{tmp_path}:{os.environ['PATH']}")
    spec = importlib.util.spec_from_file_location("bench_module_traffic", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = ["--rows", "123456", "--steps", "30", "--traffic", "measure", "--no-extras"]
    got, note = bench.measure_traffic(bench.parse_args(argv), argv)
    assert got["steps_counted"] == 4
    assert got["dominant_bytes_per_launch"] == (2 * 1000 + 100) * 1024          # FETCH x2 + WRITE, KiB -> bytes
    per_step = (2 * 1000 + 100) + (2 * 10 + 10) + (300 + 700)                   # both pre-filter launches doubled, prep as read
    assert got["step_bytes"] == per_step * 1024
    assert "counted in this run" in note
    calls = open(tmp_path / "calls.txt").read().splitlines()
    assert len(calls) == 2 and all("--pmc" in c and "trace" not in c.split(" -- ")[0] for c in calls)
    child = calls[0].split(" -- ")[1].split()
    assert child[0] == sys.executable and child[1].endswith("bench.py")
    assert "--traffic off" in calls[0] and "--traffic-child" in calls[0] and "--rows 123456" in calls[0]
    assert "--steps 3" in calls[0] and "--steps 30" not in calls[0]
