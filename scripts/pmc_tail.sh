#!/bin/bash
# Vector-memory counters of the step's kernels (what do the finaliser / prep wait for?): one --pmc pass of the bench command
# (counters only).  usage (GPU box): bash scripts/pmc_tail.sh <outdir>
# (A second pass with TCP_UTCL1_* / TCC_HIT / TCC_MISS aborted inside rocprofv3 on this pool and sat silent until it was killed:
#  those counters are left out.)
set -e
out=$1
export TMPDIR=/tmp
mkdir -p "$out"
p1="GRBM_GUI_ACTIVE TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY"
i=0
for p in "$p1"; do
  i=$((i+1))
  rocprofv3 --pmc $p --output-format csv -d "$out/t$i" -- python3 bench.py --no-extras --no-cpu-baseline --traffic off --steps 8 --warmup 2 > "$out/t$i.log" 2>&1
  echo "pass $i done"
done
python3 - "$out" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/t*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "sknnr" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    g = v.get("GRBM_GUI_ACTIVE", 0.0)
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:34s} {x:16.4g}   per GRBM cycle {x / g if g else 0:10.4g}")
PY
