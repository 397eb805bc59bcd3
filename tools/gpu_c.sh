#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 120 tools/gemm_bench lay > gpurun_out/r2c_gemm_layouts.log 2>&1; cat gpurun_out/r2c_gemm_layouts.log
for cfg in "1 1 1" "1 0 1" "0 0 1" "0 1 1" "0 0 0"; do
  set -- $cfg
  VILA_SFT_CM=$1 VILA_SFT_SIDE=$2 VILA_SFT_OPT_STREAM=$3 timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cm side opt = $cfg ->', d['ms_per_step'], 'ms  loss', d['loss'])"
done 2>&1 | tee gpurun_out/r2c_sft_variants.log
