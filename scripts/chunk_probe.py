#!/usr/bin/env python
"""Launch granularity of one call (GPU box): the bench workload with SKNNR_CHUNK_ROWS = 4M (default) .. 16M.
Each setting runs in its own process (the library reads the variable once).
usage: python scripts/chunk_probe.py"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rows in (1 << 22, 5_000_000, 1 << 23, 10_000_000 + 6144):
    env = dict(os.environ, SKNNR_CHUNK_ROWS=str(rows))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-extras", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
    j = json.loads(out)
    print(f"SKNNR_CHUNK_ROWS={rows}: {j['value']:.1f} Mq/s, {j['ms_per_step']:.2f} ms/step, pre-filter "
          f"{j['roofline']['kernel_ms_per_step']:.2f} ms, all kernels {j['roofline']['all_kernels_ms_per_step']:.2f} ms", flush=True)
