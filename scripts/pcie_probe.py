#!/usr/bin/env python
"""Development aid: host <-> device copy rates of the box (pinned memory, 1 GiB), alone and both directions at once."""
import time

import torch

n = 1 << 27  # 1 GiB of float64
h_in = torch.empty(n, dtype=torch.float64).pin_memory()
h_out = torch.empty(n, dtype=torch.float64).pin_memory()
d_a = torch.empty(n, dtype=torch.float64, device="cuda")
d_b = torch.zeros(n, dtype=torch.float64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, fn in (("H2D", lambda: d_a.copy_(h_in, non_blocking=True)), ("D2H", lambda: h_out.copy_(d_b, non_blocking=True))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {8 * n / dt / 1e9:.1f} GB/s")
t0 = time.perf_counter()
for _ in range(3):
    with torch.cuda.stream(s1):
        d_a.copy_(h_in, non_blocking=True)
    with torch.cuda.stream(s2):
        h_out.copy_(d_b, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"both at once: {8 * n / dt / 1e9:.1f} GB/s each way")
import numpy as np
a = np.ones(n // 2); b = np.empty(n // 2)
t0 = time.perf_counter(); np.copyto(b, a); dt = time.perf_counter() - t0
print(f"single-thread host memcpy (0.5 GiB): {4 * n / dt / 1e9:.1f} GB/s")
