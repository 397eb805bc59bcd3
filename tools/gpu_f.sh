#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_sampling.py tests/test_w8a8.py -m gpu -q -s 2>&1 | tail -60 > gpurun_out/r2f_pytest.log
grep -E "passed|failed|FAILED|8B-width|^E  " gpurun_out/r2f_pytest.log | head -30
for c in 0 1; do
  VILA_SFT_C_ABI=$c timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c_abi=$c ->', d['ms_per_step'], 'ms  loss', d['loss'])"
done 2>&1 | tee gpurun_out/r2f_sft_cabi.log
