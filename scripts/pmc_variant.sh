#!/bin/bash
# Development aid: SQ counters of the pre-filter kernel for one library variant (scripts/coarse_counters.py workload).
# usage (GPU box): bash scripts/pmc_variant.sh <library .so> <outdir> [rows] [k]
set -e
lib=$1; out=$2; rows=${3:-4194304}; k=${4:-5}
export TMPDIR=/tmp
export SKNNR_HIP_LIBRARY=$PWD/$lib
mkdir -p "$out"
p1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
p2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
i=0
for p in "$p1" "$p2"; do
  i=$((i+1))
  rocprofv3 --pmc $p --output-format csv -d "$out/pmc$i" -- python3 scripts/coarse_counters.py "$rows" "$k" > "$out/pmc$i.log" 2>&1
done
python3 scripts/pmc_variant_summary.py "$out" "$rows"
