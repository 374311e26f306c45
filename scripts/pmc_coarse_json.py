#!/usr/bin/env python
"""profiles/<round>_coarse_pmc.json from the --pmc passes of scripts/pmc_passes.sh and a
--kernel-trace run: per-launch averages over the largest-grid launches of the pre-filter kernel.

usage: scripts/pmc_coarse_json.py <pmc dir> <kernel_trace.csv> <rows of the largest launch, 0 = its thread count> > out.json
FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads; MI355X_MICROARCH.md, HBM
section); rocprofv3 prints FETCH_SIZE / WRITE_SIZE in KiB."""
import collections
import csv
import glob
import json
import sys

import hashlib
import os

pmc_dir, trace_csv, rows = sys.argv[1], sys.argv[2], int(sys.argv[3])
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sknnr_amd", "csrc", "libsknnr_hip.so")
vals = collections.defaultdict(list)
grid_max = 0
recs = []
for f in glob.glob(pmc_dir + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "coarse" in r["Kernel_Name"] and "matrix" not in r["Kernel_Name"]:
            recs.append(r)
            grid_max = max(grid_max, int(r["Grid_Size"]))
name = None
for r in recs:
    if int(r["Grid_Size"]) == grid_max:
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        name = r["Kernel_Name"].split("(")[0]
avg = {k: sum(v) / len(v) for k, v in vals.items()}
if rows == 0:  # coarse2_kernel: one query row per thread of the launch (1024-thread workgroups sweep 1024 rows)
    rows = grid_max
durs = []
for r in csv.DictReader(open(trace_csv)):
    if "coarse" in r["Kernel_Name"] and "matrix" not in r["Kernel_Name"]:
        durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
big = [d for d in durs if d > 0.8 * max(durs)]
ms = sum(big) / len(big) / 1e6
fetch_kib = 2.0 * avg["FETCH_SIZE"]
write_kib = avg["WRITE_SIZE"]
hbm = (fetch_kib + write_kib) * 1024.0
xcd_cycles = avg["GRBM_GUI_ACTIVE"] / 8.0
simd_cycles = xcd_cycles * 1024.0
# the whole step: FETCH_SIZE / WRITE_SIZE of every sknnr kernel summed over the run, per pre-filter bulk launch (= per step).
# FETCH_SIZE is doubled only where the reads are wide coalesced streams (the pre-filter's LDS-DMA stage copies; gfx950
# counts those at half, MI355X_MICROARCH.md "HBM"); thread-per-row and gather kernels (prep, finalize, scan) are taken as read.
by_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
n_steps = 0
seen_bulk = set()
for f in glob.glob(pmc_dir + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "sknnr" not in kn or r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        short = kn.split("(")[0].replace("void ", "")
        by_kernel[short][r["Counter_Name"]] += float(r["Counter_Value"])
        if "coarse" in kn and int(r["Grid_Size"]) == grid_max and r["Counter_Name"] == "FETCH_SIZE":
            seen_bulk.add((f, r["Dispatch_Id"]))
n_steps = max(1, len(seen_bulk))
step_by_kernel = {}
step_bytes = 0.0
for kn, cs in by_kernel.items():
    corr = 2.0 if "coarse" in kn else 1.0
    b = (corr * cs.get("FETCH_SIZE", 0.0) + cs.get("WRITE_SIZE", 0.0)) * 1024.0 / n_steps
    step_by_kernel[kn] = {"fetch_KiB_raw": cs.get("FETCH_SIZE", 0.0) / n_steps, "write_KiB": cs.get("WRITE_SIZE", 0.0) / n_steps,
                          "fetch_correction": corr, "hbm_bytes_per_step": b}
    step_bytes += b
try:
    lib_sha16 = hashlib.sha256(open(LIB, "rb").read()).hexdigest()[:16]
except OSError:
    lib_sha16 = None
out = {
    "lib_sha16": lib_sha16,
    "steps_in_the_profiled_run": n_steps,
    "step_hbm_bytes": step_bytes,
    "step_hbm_bytes_per_query_row": step_bytes / 10_000_000.0,
    "step_by_kernel": step_by_kernel,
    "_source": "rocprofv3 --pmc passes (scripts/pmc_passes.sh) and --kernel-trace of `python3 bench.py` on MI355X; "
               "per-launch averages over the largest launches of " + str(name),
    "_corrections": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads); KiB units as rocprofv3 prints them",
    "rows_per_launch": rows,
    "FETCH_SIZE_KiB_corrected": fetch_kib,
    "WRITE_SIZE_KiB": write_kib,
    "hbm_bytes_per_launch": hbm,
    "hbm_bytes_per_query_row": hbm / rows,
    "kernel_trace_avg_ms_per_launch": ms,
    "clock_GHz_from_GRBM_GUI_ACTIVE": xcd_cycles / (ms * 1e-3) / 1e9,
    "mfma_pipe_utilisation": avg["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
    "valu_issue_utilisation": 4.0 * avg["SQ_ACTIVE_INST_VALU"] / simd_cycles,
    "mfma_valu_coexec_fraction": avg["SQ_VALU_MFMA_COEXEC_CYCLES"] / simd_cycles,
    "lds_busy_fraction": avg["SQ_LDS_IDX_ACTIVE"] / (xcd_cycles * 256.0),
}
for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
          "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"):
    out[k] = avg.get(k)
print(json.dumps(out, indent=1))
