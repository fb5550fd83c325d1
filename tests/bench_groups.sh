#!/bin/bash
# ad hoc: 16- vs 64-row candidate groups at a rank-sized shard
for G in 16 64; do
  for extra in "--profile-all" ""; do
    HBMRAG_GROUP_ROWS=$G timeout -k 10 300 python bench.py --rows ${ROWS:-1250000} --batch ${BATCH:-128} --steps 100 --warmup 10 --no-latency --no-cpu-baseline $extra 2>/dev/null > /tmp/bg.json || exit 1
    python -c "
import json;d=json.load(open('/tmp/bg.json'));print('group_rows=$G', round(d['value']), round(d['ms_per_step'],3), d['kernel_ms'])"
  done
done
