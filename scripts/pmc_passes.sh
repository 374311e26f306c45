#!/bin/bash
# Separate --pmc passes (no trace domains mixed in), per MI355X_MICROARCH.md "rocprofv3 PMC slots".
# usage: scripts/pmc_passes.sh <outdir> <bench args...>
set -e
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
p1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
p2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
p3="FETCH_SIZE"
p4="WRITE_SIZE"
i=0
for p in "$p1" "$p2" "$p3" "$p4"; do
  i=$((i+1))
  rocprofv3 --pmc $p --output-format csv -d "$out/pmc$i" -- python3 bench.py "$@" > "$out/pmc$i.log" 2>&1
  echo "pass $i done"
done
