#!/usr/bin/env python
"""Development aid: event counts of the pre-filter sweep.  Needs a library built with
-DSKNNR_COARSE_COUNTERS (python -c "from sknnr_amd import _build; _build.build(True, True,
['-DSKNNR_COARSE_COUNTERS'])"); the counts are printed to stderr by sknnr_get_stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N  # noqa: E402
from sknnr_amd import synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n_ref, d = 50_000, 32
x_ref, y, _ = synth.make_problem(n_ref, 16, d, t=2)
g = torch.Generator(device="cuda").manual_seed(1)
xq = torch.randn((nq, d), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(
    synth.mixing_matrix(d), device="cuda")
ix = N.Index(x_ref, y)
o = ix.make_opts(k)
dist = torch.empty((nq, k), dtype=torch.float64, device="cuda")
idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
ix.kneighbors_device(xq.data_ptr(), nq, o, dist.data_ptr(), idx.data_ptr())
torch.cuda.synchronize()
print(ix.stats())
