#!/usr/bin/env python
"""Development aid: pre-filter timings for feature widths 48..128 (GPU box): python scripts/wide_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import config_probe as cp
cp.run("d=64 k=5", 4_000_000, 50_000, 64, 5, reps=2)
cp.run("d=48 k=5", 4_000_000, 50_000, 48, 5, reps=2)
cp.run("d=64 k=7", 2_000_000, 50_000, 64, 7, reps=2)
cp.run("d=100 k=5", 2_000_000, 50_000, 100, 5, reps=2)
cp.run("d=80 k=5", 2_000_000, 50_000, 80, 5, reps=2)
cp.run("d=96 k=5", 2_000_000, 50_000, 96, 5, reps=2)
cp.run("d=128 k=5", 1_000_000, 50_000, 128, 5, reps=2)
