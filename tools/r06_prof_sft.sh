#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for i in 1 2; do timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sft', d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_sft_r06 -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode sft --steps 4 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $O/prof_sft_r06 -name "*.db" | head -1) $O/r06_sft_step_kernel_stats.csv
grep -i "colpart\|colsum\|scatter_add\|ordered_sum\|norm_bwd\|ce_kernel" $O/r06_sft_step_kernel_stats.csv | cut -c1-150
