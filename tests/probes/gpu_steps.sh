#!/bin/bash
# Helper for gpurun calls: run steps one after the other, each under its own time limit; a step that is killed at its limit
# (or by a signal) ends the call — no further GPU step is started after a hang.  Usage: source this, then  step SECONDS cmd...
step() {
    local limit=$1; shift
    echo "=== $(date +%H:%M:%S) step (limit ${limit}s): $*"
    timeout -k 10 "$limit" "$@"
    local rc=$?
    echo "=== rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then
        echo "=== step killed (rc=$rc): stopping here"
        exit 1
    fi
    return 0
}
