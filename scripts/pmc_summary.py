#!/usr/bin/env python
"""Sum rocprofv3 --pmc counter CSVs per kernel: scripts/pmc_summary.py <dir with pmc*/> """
import csv, glob, re, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add((f, r["Dispatch_Id"]))
for name, cs in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if not name.startswith(("void sknnr", "sknnr")):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {v:.6g}")
