#!/bin/bash
# Development aid: scripts/config_quick.py cases under each dev_*.so variant.  usage (GPU box): bash scripts/variant_quick.sh <out file> <cases>
out=$1; cases=$2
for lib in sknnr_amd/csrc/dev_*.so; do
  echo "== $lib" >> "$out"
  SKNNR_HIP_LIBRARY=$PWD/$lib timeout -k 10 300 python scripts/config_quick.py "$cases" 2>/dev/null | cut -c1-170 >> "$out" || exit 1
done
