#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_full_depth.py tests/test_gpu_run.py -m gpu -q -x 2>&1 | grep "passed\|failed\|Error" | tail -4
for i in 1 2; do timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sft', d['ms_per_step'], d['loss'])"; done
