#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command.  usage: tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/ (scratch) — copy the *_kernel_stats.csv into profiles/ to have it judged.
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o trace -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_stdout.log" 2> "$OUT/bench_stderr.log"
echo "rocprofv3 rc=$?"
find "$OUT" -name "*kernel_stats*" | head
