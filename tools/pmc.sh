#!/bin/bash
# HBM traffic counters for the bench command, one counter per pass (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  usage: tools/pmc.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  OUT=$REPO/gpurun_out/pmc_${TAG}_$CTR
  mkdir -p "$OUT"
  rocprofv3 --pmc $CTR --kernel-trace -d "$OUT" -o pmc -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/stdout.log" 2> "$OUT/stderr.log"
  echo "$CTR rc=$?"
done
