#!/bin/bash
# ad-hoc sweep helper (not a test): rows x group size
for rows in "$@"; do for gr in 16 64; do
HBMRAG_GROUP_ROWS=$gr timeout -k 10 300 python bench.py --rows $rows --steps 30 --warmup 6 --profile-all --no-cpu-baseline --no-latency 2>/dev/null | GR=$gr python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('rows', d['config']['rows'], 'GR', os.environ['GR'], round(d['value']), round(d['ms_per_step'],4), d['kernel_ms'], d['all_lists_proven_exact'])"
done; done
