#!/bin/bash
# Development aid: time the pre-filter of each dev_*.so variant (scripts/coarse_counters.py workload).
# usage (GPU box): bash scripts/variant_times.sh <out file> [rows] [k ...]
out=$1; rows=${2:-4194304}; shift; shift
ks=${@:-5}
for lib in sknnr_amd/csrc/dev_*.so; do
  for k in $ks; do
    for rep in 1 2; do
      echo "== $lib k=$k" >> "$out"
      SKNNR_HIP_LIBRARY=$PWD/$lib timeout -k 10 120 python scripts/coarse_counters.py "$rows" "$k" 2>/dev/null | tail -1 >> "$out" || exit 1
    done
  done
done
