#!/bin/bash
# Development aid: the benchmark workload (10M x 50k x 32, cells on) under each dev_*.so variant, twice.
# usage (GPU box): bash scripts/variant_cells.sh <out file> [k]
out=$1; k=${2:-5}
for rep in 1 2; do
  for lib in sknnr_amd/csrc/dev_*.so; do
    echo "== $lib k=$k" >> "$out"
    SKNNR_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python scripts/cells_probe.py 10000000 "$k" 6 2>/dev/null | cut -c1-200 >> "$out" || exit 1
  done
done
