#!/bin/bash
# HBM traffic of the contraction-major GEMMs on the four SFT wgrad shapes: one rocprofv3 --pmc pass per
# (shape, counter), each over `tools/gemm_bench one <idx>` = 23 calls of ONE GEMM (3 warm-up + 20 timed) with the default kernels and launch
# policies.  Writes gpurun_out/pmc_gemm_sft/summary.{txt,json}.   usage: bash tools/pmc_gemm_sft.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/pmc_gemm_sft; mkdir -p "$O"
[ -x "$REPO/tools/gemm_bench" ] || (cd "$REPO" && python -c "from vila_amd import build as b; b.build(); b.build_tools()") || { echo "tools/gemm_bench missing and not buildable"; exit 2; }
cd /tmp && export TMPDIR=/tmp
for IDX in 5 6 7 8; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    D=$O/${IDX}_$CTR; mkdir -p "$D"
    timeout 300 rocprofv3 --pmc $CTR --kernel-trace -d "$D" -o pmc -- "$REPO/tools/gemm_bench" one $IDX > "$D/stdout.log" 2> "$D/stderr.log"
  done
done
cd "$REPO" && python tools/pmc_gemm_summary.py "$O" | tee "$O/summary.txt"
find "$O" -name "*.db" -delete
