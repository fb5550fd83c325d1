#!/bin/bash
# ad hoc: step time vs batches in flight, one box
for F in ${FLIGHTS:-1 2 3 4}; do
  timeout -k 10 300 python bench.py --in-flight $F --steps 50 --warmup 6 --no-latency --no-cpu-baseline 2>/dev/null > /tmp/if.json || exit 1
  python -c "
import json;d=json.load(open('/tmp/if.json'));print('in-flight=$F', round(d['value']), round(d['ms_per_step'],3), d['kernel_ms'], d['host_enqueue_ms_per_step'])"
done
