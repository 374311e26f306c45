#!/bin/bash
# Development aid (one GPU): one rank's share of the 10M-row job at N = 2 / 4 / 8, as ONE call and as the chunked,
# gather-overlapped step the N > 1 bench runs (--force-dist: a one-rank RCCL group, the all-gathers are local copies).
# usage (GPU box): bash scripts/rank_share_probe.sh <out file>
out=$1
for n in 2 4 8; do
  rows=$((10000000 / n))
  for mode in "" "--force-dist --gather-chunks 2" "--force-dist --gather-chunks 3" "--force-dist --gather-chunks 4"; do
    python bench.py --rows $rows $mode --no-extras --no-cpu-baseline --steps 20 2>/dev/null > /tmp/rs.json || exit 1
    python - "$n" "$mode" >> "$out" <<'PY'
import json, sys
r = json.load(open("/tmp/rs.json"))
print(f"N={sys.argv[1]} share {r['config']['rows_per_gpu']} rows {(sys.argv[2][-15:] + ' + gather (1-rank group)') if sys.argv[2] else 'one call':42s}: "
      f"{r['ms_per_step']:.3f} ms/step, kernels {r['roofline']['all_kernels_ms_per_step']:.3f} ms, pre-filter {r['roofline']['kernel_ms_per_step']:.3f} ms "
      f"-> {10e6 / r['ms_per_step'] / 1e3:.0f} Mq/s for the job at that N")
PY
  done
done
