#!/bin/bash
# Collects the round's evidence on a GPU box in ONE call (outputs under gpurun_out/<tag>_*; copy the summaries into profiles/):
#   bash profiles/collect_round.sh r4
# 1. default bench, un-profiled (CPU baseline, parity gate, latencies, api_concurrent, config4_full)
# 2. the same under rocprofv3 --kernel-trace --stats (kernel stats + the bench line of the SAME process)
# 3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, kernel-trace only)
# 4. rank-sized, 256-query, Zipf and ingest lines; encoder / cross-encoder probes
# The program after `--` is always python3 itself (no env / bash -c hop under the profiler).
set -u
tag=${1:-r4}
out=gpurun_out
mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
run() { echo "=== $(date +%H:%M:%S) $*"; timeout -k 10 "$@"; rc=$?; echo "=== rc=$rc"; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "killed: stopping"; exit 1; fi; }
run 600 python3 bench.py > $out/${tag}_final_bench.log 2> $out/${tag}_final_bench.err
run 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o ${tag} -- python3 bench.py --no-cpu-baseline --no-api-concurrent > $out/${tag}_bench_under_rocprof.log 2> $out/${tag}_bench_under_rocprof.err
run 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_f -o f -- python3 bench.py --no-cpu-baseline --no-latency --no-config4-full --steps 20 > $out/${tag}_pmc_f.log 2> $out/${tag}_pmc_f.err
run 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_w -o w -- python3 bench.py --no-cpu-baseline --no-latency --no-config4-full --steps 20 > $out/${tag}_pmc_w.log 2> $out/${tag}_pmc_w.err
run 300 python3 bench.py --rows 1250000 --steps 300 --simulate-ranks 8 --profile-all --no-cpu-baseline --no-latency > $out/${tag}_rank_sized_bench.log 2> $out/${tag}_rank_sized_bench.err
run 300 python3 bench.py --batch 256 --no-cpu-baseline --no-api-concurrent --no-config4-full --no-latency > $out/${tag}_batch256_bench.log 2> $out/${tag}_batch256_bench.err
run 300 python3 bench.py --batch 256 --dense-kernel-mask 16 --no-cpu-baseline --no-api-concurrent --no-config4-full --no-latency > $out/${tag}_batch256_q64_bench.log 2> $out/${tag}_batch256_q64_bench.err
run 300 python3 bench.py --sparse-dist zipf --no-cpu-baseline --no-api-concurrent --no-config4-full > $out/${tag}_zipf_bench.log 2> $out/${tag}_zipf_bench.err
run 400 python3 bench.py --ingest --ingest-docs 1024 > $out/${tag}_ingest_bench.log 2> $out/${tag}_ingest_bench.err
run 200 python3 tests/perf_probe_ce.py 128 > $out/${tag}_ce_probe_128.log 2> $out/${tag}_ce_probe_128.err
run 200 python3 tests/perf_probe_ce.py 512 > $out/${tag}_ce_probe_512.log 2> $out/${tag}_ce_probe_512.err
PROBE_PROFILE=1 run 200 python3 tests/perf_probe_encoder.py 512 1024 768 12 > $out/${tag}_encoder_probe.log 2> $out/${tag}_encoder_probe.err
run 100 tests/probes/bin/el_probe_stamp 327680 10 1 > $out/${tag}_el_stamp.log 2>&1
for ab in 0 1 2 4 8 15; do run 60 tests/probes/bin/el_probe_$ab 327680 10 1; done > $out/${tag}_el_ablate.log 2>&1
echo "=== done"
