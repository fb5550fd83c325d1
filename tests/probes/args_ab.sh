#!/bin/bash
# Same-box comparison of bench argument sets (alternating, two rounds): tests/probes/args_ab.sh "common args" "args A" "args B" ...
source tests/probes/gpu_steps.sh
COMMON=$1; shift
for round in 1 2; do
  for extra in "$@"; do
    step 400 python bench.py --no-cpu-baseline --no-api-concurrent --no-config4-full --no-latency $COMMON $extra > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python - "$extra" <<'PY'
import json, sys
d = [json.loads(l) for l in open("gpurun_out/ab_tmp.json") if l.startswith("{")][0]
print(repr(sys.argv[1]), round(d["value"]), round(d["ms_per_step"], 4), d["kernel_ms"], d.get("finishing_alone", {}).get("finish_us"), d["all_lists_proven_exact"])
PY
  done
done
