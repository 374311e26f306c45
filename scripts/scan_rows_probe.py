#!/usr/bin/env python
"""Development aid (GPU box): the float64 scan alone (SKNNR_EXACT_ONLY=1) on 50k x 32 reference rows, k = 5, for several
call sizes.  usage: [SKNNR_HIP_LIBRARY=variant.so] python scripts/scan_rows_probe.py"""
import os
import sys

os.environ["SKNNR_EXACT_ONLY"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

x_ref, _, x_q = synth.make_problem(50_000, 200_000, 32, t=1)
ix = N.Index(x_ref)
xq = torch.as_tensor(x_q, device="cuda")
out = []
for nq in (500, 2_000, 8_700, 30_000, 200_000):
    o = ix.make_opts(5)
    dist = torch.empty((nq, 5), dtype=torch.float64, device="cuda")
    idx = torch.empty((nq, 5), dtype=torch.int64, device="cuda")
    for rep in range(3):
        ix.reset_stats()
        ix.kneighbors_device(xq.data_ptr(), nq, o, dist.data_ptr(), idx.data_ptr())
        torch.cuda.synchronize()
        st = ix.stats()
    out.append(f"{nq}: {st['last_kernel_ms']:.2f} ms")
print("scan alone, rows: " + "; ".join(out) + "  [" + os.path.basename(os.environ.get("SKNNR_HIP_LIBRARY", "in-tree")) + "]", flush=True)
