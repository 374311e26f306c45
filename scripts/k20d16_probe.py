#!/usr/bin/env python
"""Development aid: one configuration (k = 20, 16 features, 2M rows) for kernel traces: python scripts/k20d16_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import config_probe as cp
cp.run("k=20 d=16", 2_000_000, 50_000, 16, 20, reps=2)
