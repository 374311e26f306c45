#!/usr/bin/env python
"""Development aid: BASELINE configs 2-5 and the two hostile laws, one pass each (bench.extra_config), printed compactly."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sknnr_amd import synth  # noqa: E402

which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "c3", "c4", "c5", "cube", "sorted"]
cfgs = {
    "c2": lambda: bench.extra_config("C2", "euclidean", 1_000_000, 10_000, 16, 5, torch, 0),
    "c3": lambda: bench.extra_config("C3", "gnn", 10_000_000, 50_000, 32, 7, torch, 0, predict="distance"),
    "c4": lambda: bench.extra_config("C4", "mahalanobis", 10_000_000, 50_000, 64, 5, torch, 0),
    "c5": lambda: bench.extra_config("C5", "msn", 6_250_000, 100_000, 32, 1, torch, 0, n_components=8, dataframe_ids=True),
    "cube": lambda: bench.extra_config("cube", "euclidean", 10_000_000, 50_000, 32, 5, torch, 0, law="uniform",
                                       x_ref=np.random.default_rng(0).random((50_000, 32))),
    "sorted": lambda: bench.extra_config("sorted", "gnn", 10_000_000, 50_000, 32, 5, torch, 0,
                                         x_ref=(lambda s: np.ascontiguousarray(s[np.argsort(s[:, 0])]))(synth.make_features(50_000, 32, seed=0))),
    "k10": lambda: bench.extra_config("k10", "gnn", 4_000_000, 50_000, 32, 10, torch, 0),
    "k7": lambda: bench.extra_config("k7", "gnn", 10_000_000, 50_000, 32, 7, torch, 0),
    "k7d64": lambda: bench.extra_config("k7d64", "mahalanobis", 4_000_000, 50_000, 64, 7, torch, 0),
    "k14": lambda: bench.extra_config("k14", "gnn", 4_000_000, 50_000, 32, 14, torch, 0),
    "k16": lambda: bench.extra_config("k16", "gnn", 2_000_000, 50_000, 32, 16, torch, 0),
    "k25": lambda: bench.extra_config("k25", "gnn", 2_000_000, 50_000, 32, 25, torch, 0),
    "k30": lambda: bench.extra_config("k30", "gnn", 2_000_000, 50_000, 32, 30, torch, 0),
    "k20": lambda: bench.extra_config("k20", "gnn", 2_000_000, 50_000, 32, 20, torch, 0),
}
for w in which:
    c = cfgs[w]()
    print(f"{w:7s} {c['Mq_s']:7.1f} Mq/s  frac {c['frac']:.3f}  pre-filter {c['prefilter_ms']:7.2f} ms  kernels {c['kernel_ms']:7.2f} ms  wall {c['ms']:7.2f} ms  "
          f"fallbacks {c['exact_fallbacks_per_pass']}  oracle {c['oracle_check']['index_rows_equal']}/{c['oracle_check']['rows']} {c['oracle_check']['dist_bit_equal']}", flush=True)
