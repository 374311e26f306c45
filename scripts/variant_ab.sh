#!/bin/bash
# Development aid: the benchmark workload (10M x 50k x 32, cells on) under the named dev_*.so variants, interleaved, `reps` times.
# usage (GPU box): bash scripts/variant_ab.sh <out file> <reps> <variant> [<variant> ...]      (variant = file stem under sknnr_amd/csrc/)
out=$1; reps=$2; shift 2
for rep in $(seq 1 "$reps"); do
  for v in "$@"; do
    echo "== $v" >> "$out"
    SKNNR_HIP_LIBRARY=$PWD/sknnr_amd/csrc/$v.so timeout -k 10 200 python scripts/cells_probe.py 10000000 5 6 2>>"$out.err" | cut -c1-260 >> "$out" || exit 1
  done
done
