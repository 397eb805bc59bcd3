#!/bin/bash
# attention backward on two streams: parity + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -5 > "$O/r2o_pytest_train.log"; tail -2 "$O/r2o_pytest_train.log"
for v in 0 1 0 1; do
  VILA_SFT_ATTN_STREAM=$v timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/r2o_sft_$v.err" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('attn_stream=$v ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"
done 2>&1 | tee "$O/r2o_sft_attn_stream.log"
timeout 600 bash tools/profile.sh r2o_sft --mode sft --steps 3 --warmup 1 2>&1 | tail -1
if [ -f "$O/prof_r2o_sft/trace_results.db" ]; then
  python tools/rocpd_timeline.py "$O/prof_r2o_sft/trace_results.db" im2col_kernel @next -2 "$O/r2o_sft_step_timeline.txt"; head -3 "$O/r2o_sft_step_timeline.txt"
  python tools/rocpd_summary.py "$O/prof_r2o_sft/trace_results.db" "$O/r2o_sft_kernel_stats.csv"
  rm -f "$O/prof_r2o_sft/trace_results.db"
fi
