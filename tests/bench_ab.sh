#!/bin/bash
# ad hoc A/B of an environment switch inside one box: bench_ab.sh VAR [batch]
VAR=$1; B=${2:-128}
for rep in 1 2; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $VAR=1; else unset $VAR; fi
    timeout -k 10 300 python bench.py --batch $B --steps 50 --warmup 6 --no-latency --no-cpu-baseline 2>/dev/null > /tmp/ab.json || exit 1
    python -c "
import json;d=json.load(open('/tmp/ab.json'));print('$VAR=$on', 'B=$B', round(d['value']), round(d['ms_per_step'],3), d['kernel_ms'], d['all_lists_proven_exact'])"
  done
done
