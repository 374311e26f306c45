#!/usr/bin/env python
"""Development aid: pre-filter time and oracle parity for one shape.  usage: shape_probe.py n_ref nq d k [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

n_ref, nq, d, k = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
x_ref = synth.make_features(n_ref, d, seed=0)
g = torch.Generator(device="cuda").manual_seed(1)
xq = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(synth.mixing_matrix(d), device="cuda")
ix = N.Index(x_ref)
o = ix.make_opts(k)
dist = torch.empty((nq, k), dtype=torch.float64, device="cuda")
idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
for _ in range(reps):
    ix.reset_stats()
    ix.kneighbors_device(xq.data_ptr(), nq, o, dist.data_ptr(), idx.data_ptr())
    torch.cuda.synchronize()
    st = ix.stats()
n_chk = min(nq, 4096)
od, oi = O.kneighbors(x_ref, xq[:n_chk].cpu().numpy(), k, "expanded")
ok = np.array_equal(idx[:n_chk].cpu().numpy(), oi) and np.array_equal(dist[:n_chk].cpu().numpy(), od)
print(f"V2={os.environ.get('SKNNR_COARSE_V2', '1')} {n_ref}x{nq}x{d} k={k}: pre-filter {st['total_coarse_ms']:.2f} ms, all kernels "
      f"{st['total_kernel_ms']:.2f} ms, fallbacks {st['exact_fallbacks']}, {nq / st['total_kernel_ms'] / 1e3:.1f} Mq/s, oracle slice equal: {ok}")
