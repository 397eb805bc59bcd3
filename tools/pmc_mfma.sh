#!/bin/bash
# MFMA-pipe utilisation counters (one SQ pass) for a bench command.  usage: tools/pmc_mfma.sh <tag> [bench args...]
#   MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)   (tools/pmc_mfma_summary.py)
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_mfma_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d "$OUT" -o pmc -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/stdout.log" 2> "$OUT/stderr.log"
echo "rc=$?"
python "$REPO/tools/pmc_mfma_summary.py" "$OUT"/pmc_results.db > "$OUT/summary.txt" 2>&1
rm -f "$OUT"/pmc_results.db          # keep gpurun_out small: only the summary travels back
head -16 "$OUT/summary.txt"
