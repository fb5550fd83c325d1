#!/bin/bash
# Run a list of GPU steps one after the other on a gpurun box; a step that TIMES OUT (or is killed) ends the batch —
# a plain failure does not.  Usage: run_gpu_batch.sh <name> <timeout_s> <cmd...> ::: <name> <timeout_s> <cmd...> ::: ... (":::" separates the steps: "--" belongs to rocprofv3)
# Output of each step goes to gpurun_out/<name>.log
set -u
mkdir -p gpurun_out
while [ $# -gt 0 ]; do
    name=$1; t=$2; shift 2
    cmd=()
    while [ $# -gt 0 ] && [ "$1" != ":::" ]; do cmd+=("$1"); shift; done
    [ $# -gt 0 ] && shift
    echo "=== $name: ${cmd[*]}"
    timeout -k 10 "$t" "${cmd[@]}" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
    rc=$?
    echo "=== $name rc=$rc"
    tail -n 3 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "=== $name timed out / killed: stopping the batch"
        tail -n 20 "gpurun_out/$name.err"
        exit $rc
    fi
done
exit 0
