#!/usr/bin/env python
"""Development aid: reference rows handed over SORTED by their first feature (a caller's order that
correlates with the data): the per-index choice of the image order against the caller's order."""
import os
import subprocess
import sys

if len(sys.argv) > 1:  # child: one measurement
    import time

    import numpy as np
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as O
    from sknnr_amd import _native as N
    from sknnr_amd import synth

    d = int(sys.argv[1])
    n_ref, nq, k = 50_000, 4_194_304, 5
    x_ref = synth.make_features(n_ref, d, seed=0)
    x_ref = x_ref[np.argsort(x_ref[:, 0])]
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(synth.mixing_matrix(d), device="cuda")
    ix = N.Index(x_ref)
    o = ix.make_opts(k)
    dist = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.kneighbors_device(q.data_ptr(), nq, o, dist.data_ptr(), idx.data_ptr())
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    od, oi = O.kneighbors(x_ref, q[:1000].cpu().numpy(), k, "expanded")
    print(f"d={d} SKNNR_IMAGE_ORDER={os.environ.get('SKNNR_IMAGE_ORDER', 'auto')}: {best * 1e3:.1f} ms -> {nq / best / 1e6:.1f} Mq/s, "
          f"slice bad rows {int((idx[:1000].cpu().numpy() != oi).any(axis=1).sum())}", flush=True)
else:
    for d in ("8", "32"):
        for order in ("0", None):
            env = dict(os.environ)
            env.pop("SKNNR_IMAGE_ORDER", None)
            if order is not None:
                env["SKNNR_IMAGE_ORDER"] = order
            subprocess.run([sys.executable, os.path.abspath(__file__), d], env=env, check=True)
