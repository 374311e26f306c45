#!/bin/bash
# Round profile of `python3 bench.py` on the GPU box: kernel trace + stats, then the --pmc passes (no trace
# domains mixed in).  Outputs under gpurun_out/<tag>/; copy the summaries into profiles/.
# usage: bash scripts/profile_round.sh <tag>
set -e
tag=$1
out=gpurun_out/$tag
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-extras --no-cpu-baseline --traffic off > "$out/trace.log" 2>&1
echo "trace done"
bash scripts/pmc_passes.sh "$out" --no-extras --no-cpu-baseline --traffic off
tr=$(find "$out/trace" -name "*kernel_trace.csv" | head -1)
st=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
python3 scripts/trim_stats.py "$st" "$out/bench_kernel_stats.csv"  # (the torch RNG kernel's name runs to kilobytes)
python3 scripts/pmc_coarse_json.py "$out" "$tr" 0 > "$out/coarse_pmc.json"
cat "$out/coarse_pmc.json"
head -8 "$out/bench_kernel_stats.csv"
