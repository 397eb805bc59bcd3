#!/bin/bash
# round-2 GPU pass B: contraction-major GEMMs in the trainer, wgrad side stream, per-bucket AdamW
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -m gpu -q -s 2>&1 | tail -40 > gpurun_out/r2b_pytest.log
tail -6 gpurun_out/r2b_pytest.log
timeout 600 python bench.py --mode sft --steps 4 --warmup 2 > gpurun_out/r2b_sft.json 2> gpurun_out/r2b_sft.err
echo "sft rc=$?"; cat gpurun_out/r2b_sft.json | cut -c1-400
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_sft -o sft -- python "$GRAFT_REPO_ROOT/bench.py" --mode sft --steps 4 --warmup 1 > /tmp/prof_sft.log 2>&1
echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/prof_sft -name "*kernel_stats.csv" | head -1); echo "stats: $f"
[ -n "$f" ] && cp "$f" gpurun_out/r2b_sft_kernel_stats.csv && head -25 gpurun_out/r2b_sft_kernel_stats.csv | cut -c1-150
