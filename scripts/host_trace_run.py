import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sknnr_amd import _native as N
from sknnr_amd import synth
nq, k = 10_000_000, 5
x_ref, y, _ = synth.make_problem(50_000, 16, 32, t=8)
g = torch.Generator(device="cuda").manual_seed(1)
xq = (torch.randn((nq, 32), dtype=torch.float64, device="cuda", generator=g) @ torch.tensor(synth.mixing_matrix(32), device="cuda")).cpu().numpy()
ix = N.Index(x_ref, y)
o = ix.make_opts(k)
d_out = np.zeros((nq, k)); i_out = np.zeros((nq, k), dtype=np.int64)
for rep in range(3):
    t0 = time.perf_counter()
    with ix.open_stream(o) as s:
        for a in range(0, nq, 1_000_000):
            s.push(xq[a:a + 1_000_000], out_idx=i_out[a:a + 1_000_000], out_dist=d_out[a:a + 1_000_000])
    print("stream", (time.perf_counter() - t0) * 1e3, "ms", flush=True)
