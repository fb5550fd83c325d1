#!/bin/bash
# config4_full with the forward exclusive (default) and overlapped, one process each, same box
source tests/probes/gpu_steps.sh
for flag in "" "--ce-overlap" "" "--ce-overlap"; do
  step 500 python bench.py --no-cpu-baseline --no-api-concurrent --no-latency --steps 20 $flag > gpurun_out/ce_tmp.json 2> gpurun_out/ce_tmp.err
  python - "$flag" <<'PY'
import json, sys
d = [json.loads(l) for l in open("gpurun_out/ce_tmp.json") if l.startswith("{")][0]
c = d["config4_full"]
for T in ("seq_len_128", "seq_len_512"):
    x = c[T]; ce = x["cross_encoder"]
    print(sys.argv[1] or "exclusive", T, round(x["value"]), "q/s", round(x["ms_per_step"], 3), "ms/step  forward", round(ce["forward_ms_per_step"], 3), "ms frac", round(ce["frac"], 4), "alone", round(ce.get("forward_alone_ms", 0), 3), round(ce.get("frac_alone", 0), 4), "flags", d["all_lists_proven_exact"])
PY
done
