#!/bin/bash
# Same-box A/B of two builds of the library on the default bench line: tests/probes/ab_default.sh LIB_A LIB_B [extra bench args]
source tests/probes/gpu_steps.sh
A=$1; B=$2; shift 2
for round in 1 2; do
  for lib in "$A" "$B"; do
    export HBMRAG_LIB=$lib
    step 400 python bench.py --no-cpu-baseline --no-api-concurrent --no-config4-full --no-latency --steps 50 "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python - "$lib" <<'PY'
import json, sys
d = [json.loads(l) for l in open("gpurun_out/ab_tmp.json") if l.startswith("{")][0]
print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), d["kernel_ms"], d.get("finishing_alone", {}).get("finish_us"), d["all_lists_proven_exact"])
PY
  done
done
