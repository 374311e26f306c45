#!/usr/bin/env python
"""Summary of scripts/pmc_variant.sh: per-unit (32-reference tile x 32-query block) instruction counts and
SIMD-cycle shares of the pre-filter kernel (n_ref = 50,000 -> 1,563 tiles)."""
import collections
import csv
import glob
import sys

out, rows = sys.argv[1], int(sys.argv[2])
vals = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "coarse" in r["Kernel_Name"] and "matrix" not in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in vals.items()}
units = rows / 32 * 1563
simd_cycles = a["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
print(f"units {units:.3e}; SIMD-cycles per unit {simd_cycles / units:.1f}")
for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    print(f"  {k:28s} {a[k] / units:7.2f} per unit")
print(f"  MFMA pipe busy               {a['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:6.1%} of SIMD cycles")
print(f"  VALU issue (4 x ACTIVE_INST) {4 * a['SQ_ACTIVE_INST_VALU'] / simd_cycles:6.1%}")
print(f"  MFMA/VALU co-execution       {a['SQ_VALU_MFMA_COEXEC_CYCLES'] / simd_cycles:6.1%}")
print(f"  LDS busy                     {a['SQ_LDS_IDX_ACTIVE'] / (a['GRBM_GUI_ACTIVE'] / 8.0 * 256.0):6.1%} of CU cycles")
wc = a["SQ_WAVE_CYCLES"]
print(f"  wave cycles: WAIT_ANY {a['SQ_WAIT_ANY'] / wc:6.1%}  WAIT_INST_ANY {a['SQ_WAIT_INST_ANY'] / wc:6.1%}  "
      f"ACTIVE_INST_ANY {a['SQ_ACTIVE_INST_ANY'] / wc:6.1%}  WAIT_INST_LDS {a['SQ_WAIT_INST_LDS'] / wc:6.1%}")
print(f"  waves per SIMD (avg resident) {wc * 4 / simd_cycles:.2f}")
