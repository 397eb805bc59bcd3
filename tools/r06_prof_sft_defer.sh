#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
VILA_SFT_OPT_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_sft_defer -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode sft --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $O/prof_sft_defer -name "*.db" | head -1) $O/r06_sft_defer_kernel_stats.csv
find $O/prof_sft_defer -name "*.db" -delete
