#!/usr/bin/env python
"""Development aid: 8 .. 23 neighbours at 48 / 64 features (pooled lists at three and four K-steps against the first-generation
kernel: SKNNR_COARSE_V2=0): python scripts/wide_k_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import config_probe as cp
cp.run("k=10 d=64", 1_000_000, 50_000, 64, 10, reps=2)
cp.run("k=14 d=64", 1_000_000, 50_000, 64, 14, reps=2)
cp.run("k=10 d=48", 1_000_000, 50_000, 48, 10, reps=2)
cp.run("k=20 d=48", 1_000_000, 50_000, 48, 20, reps=2)
cp.run("k=25 d=48", 1_000_000, 50_000, 48, 25, reps=2)
cp.run("k=20 d=64", 1_000_000, 50_000, 64, 20, reps=2)
cp.run("k=30 d=64", 1_000_000, 50_000, 64, 30, reps=2)
