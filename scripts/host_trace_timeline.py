#!/usr/bin/env python
"""Development aid: timeline of the large copies and the bulk pre-filter launches of the last stream run in a
rocprofv3 --kernel-trace --memory-copy-trace capture of scripts/host_trace_run.py.  usage: host_trace_timeline.py <dir>"""
import csv
import glob
import sys

root = sys.argv[1]
mc = glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True)[0]
kt = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
rows = list(csv.DictReader(open(mc)))
print("copy columns:", list(rows[0].keys()))
for r in rows:
    size = r.get("Bytes") or r.get("Size") or "0"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s" % (r.get("Direction", "?"), size)))
for r in csv.DictReader(open(kt)):
    n = r["Kernel_Name"]
    if "sknnr" in n:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + n.split("(")[0].replace("void sknnr::", "")[:30]))
ev.sort()
big = [e for e in ev if e[2].startswith("K coarse2_kernel<2, 6, 16")]
t_lo = big[-10][0] - 8_000_000
for s, e, name in ev:
    if s < t_lo:
        continue
    is_big_copy = name.startswith("COPY") and name.split()[-1].isdigit() and int(name.split()[-1]) > 1_000_000
    if is_big_copy or "coarse2_kernel<2, 6, 16" in name or "finalize" in name:
        print(f"{(s - t_lo) / 1e6:8.2f} -> {(e - t_lo) / 1e6:8.2f} ms  ({(e - s) / 1e6:6.2f})  {name}")
