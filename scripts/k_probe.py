#!/usr/bin/env python
"""Development aid: throughput for n_neighbors 8..31 (16- and 32-entry lists) on the GPU box: python scripts/k_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import config_probe as cp
cp.run("k=8", 2_000_000, 50_000, 32, 8, reps=2)
cp.run("k=10", 2_000_000, 50_000, 32, 10, reps=2)
cp.run("k=15", 2_000_000, 50_000, 32, 15, reps=2)
cp.run("k=20", 1_000_000, 50_000, 32, 20, reps=2)
cp.run("k=31", 1_000_000, 50_000, 32, 31, reps=2)
cp.run("k=10 d=64", 1_000_000, 50_000, 64, 10, reps=2)
