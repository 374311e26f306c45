#!/usr/bin/env python
"""Development aid: does a concurrent host-to-device copy slow the hot path's kernels?  1M-row device-resident calls
(kernel time from the handle's events) alone and while a 1 GiB pinned H2D copy runs on another stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

nq, k = 1_000_000, 5
x_ref, y, _ = synth.make_problem(50_000, 16, 32, t=8)
g = torch.Generator(device="cuda").manual_seed(1)
xq = torch.randn((nq, 32), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(synth.mixing_matrix(32), device="cuda")
ix = N.Index(x_ref, y)
o = ix.make_opts(k)
dd = torch.empty((nq, k), dtype=torch.float64, device="cuda")
di = torch.empty((nq, k), dtype=torch.int64, device="cuda")
n = 1 << 27
h = torch.empty(n, dtype=torch.float64).pin_memory()
d = torch.empty(n, dtype=torch.float64, device="cuda")
side = torch.cuda.Stream()
def run(busy, reps=3):
    torch.cuda.synchronize()
    ix.reset_stats()
    if busy:
        with torch.cuda.stream(side):
            d.copy_(h, non_blocking=True)
    for _ in range(reps):
        ix.kneighbors_device(xq.data_ptr(), nq, o, dd.data_ptr(), di.data_ptr())
    torch.cuda.synchronize()
    st = ix.stats()
    return st["total_kernel_ms"] / reps, st["total_coarse_ms"] / reps
run(False)
for _ in range(2):
    a, b = run(False), run(True)
print(f"HSA_ENABLE_SDMA={os.environ.get('HSA_ENABLE_SDMA')}: 1M-row call alone: kernels {a[0]:.2f} ms (pre-filter {a[1]:.2f}); beside a host-to-device copy: {b[0]:.2f} ms (pre-filter {b[1]:.2f})")
