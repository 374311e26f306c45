#!/bin/bash
# Round-end check on the GPU box: the default bench line (saved), then the randomised differential test, small and
# benchmark-sized cases.  usage: bash scripts/final_check.sh [fuzz seconds]
secs=${1:-300}
mkdir -p gpurun_out/final
python bench.py > gpurun_out/r04_bench_output.json 2> gpurun_out/r04_bench_err.txt || exit 1
python scripts/bench_fields.py gpurun_out/r04_bench_output.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_bench_output.json"))
print("frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], "step traffic / algorithmic", r["roofline"]["traffic_over_algorithmic_bytes_step"],
      "host calls ms", r.get("host_to_host_calls_ms"), "lib", r["library_sha16"])
PY
timeout -k 10 $((secs + 100)) python scripts/fuzz_parity.py $secs 61 > gpurun_out/final/fuzz_small.txt 2>&1; tail -1 gpurun_out/final/fuzz_small.txt
timeout -k 10 $((secs + 100)) python scripts/fuzz_parity.py $secs 62 --big > gpurun_out/final/fuzz_big.txt 2>&1; tail -1 gpurun_out/final/fuzz_big.txt
