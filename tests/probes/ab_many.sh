#!/bin/bash
# Same-box comparison of several builds of the library on a bench line: tests/probes/ab_many.sh "bench args" LIB...
source tests/probes/gpu_steps.sh
ARGS=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    export HBMRAG_LIB=$lib
    step 400 python bench.py --no-cpu-baseline --no-api-concurrent --no-config4-full --no-latency $ARGS > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python - "$lib" <<'PY'
import json, sys
d = [json.loads(l) for l in open("gpurun_out/ab_tmp.json") if l.startswith("{")][0]
print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), d["kernel_ms"], d.get("finishing_alone", {}).get("finish_us"), d["all_lists_proven_exact"])
PY
  done
done
