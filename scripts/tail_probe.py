#!/usr/bin/env python
"""Thin-round variant of the pre-filter (GPU box): bench workload and small calls with SKNNR_COARSE_TAIL=1 / 0.
usage: python scripts/tail_probe.py"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tail in ("1", "0"):
    env = dict(os.environ, SKNNR_COARSE_TAIL=tail)
    for rows in (10_000_000, 1_250_000, 40_000, 8_000):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-extras", "--no-cpu-baseline", "--rows", str(rows)],
                             env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
        j = json.loads(out)
        print(f"SKNNR_COARSE_TAIL={tail} rows={rows}: {j['value']:.2f} Mq/s, {j['ms_per_step']:.3f} ms/step, pre-filter "
              f"{j['roofline']['kernel_ms_per_step']:.3f} ms", flush=True)
