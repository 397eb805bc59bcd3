#!/bin/bash
# Per-launch rows of one TTFT window (start, end, duration of every kernel in order).  usage (GPU box): bash tools/r06_ttft_rows.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/prof_rows -o trace -- python $REPO/bench.py --no-cpu-baseline --no-sft --no-sustain --steps 8 --warmup 2 > $O/prof_rows.log 2>&1
DB=$(find $O/prof_rows -name "*.db" | head -1)
cd $REPO && python tools/rocpd_rows.py "$DB" im2col_kernel argmax_stage2 -2 $O/r06_ttft_rows.txt "attn_fwd_kernel<128"
find $O/prof_rows -name "*.db" -delete
wc -l $O/r06_ttft_rows.txt
