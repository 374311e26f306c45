#!/usr/bin/env python
"""Development aid: 16 .. 25 neighbours (pooled lists of 12 / 16) on the GPU box: python scripts/k16_probe.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import config_probe as cp
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
for k in (16, 20, 23, 25):
    cp.run(f"k={k}", rows, 50_000, 32, k, reps=2)
cp.run("k=20 d=16", rows, 50_000, 16, 20, reps=2)
