#!/usr/bin/env python
"""Copy a rocprofv3 kernel_stats.csv with kernel names cut to 160 characters (the torch RNG kernel's name runs to kilobytes)."""
import csv
import sys

rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w", newline=""))
for r in rows:
    w.writerow([r[0][:160]] + r[1:])
