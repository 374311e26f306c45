#!/usr/bin/env python
"""Development aid: does a host-to-device copy overlap the pre-filter?  Times a 1 GiB pinned H2D copy alone and while a
10M-row device-resident kneighbors call is running on another stream."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

nq, k = 10_000_000, 5
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
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def copy_ms(busy):
    torch.cuda.synchronize()
    if busy:
        ix.kneighbors_device(xq.data_ptr(), nq, o, dd.data_ptr(), di.data_ptr())
    with torch.cuda.stream(side):
        ev0.record(side)
        d.copy_(h, non_blocking=True)
        ev1.record(side)
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1)
for _ in range(2):
    a, b = copy_ms(False), copy_ms(True)
print(f"1 GiB H2D alone: {a:.1f} ms ({8 * n / a / 1e6:.1f} GB/s); while the pre-filter runs: {b:.1f} ms ({8 * n / b / 1e6:.1f} GB/s)")
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
