#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_run.py -m gpu -q -x 2>&1 | grep "passed\|failed" | tail -3
for i in 1 2 3; do
 timeout 300 python bench.py --mode sft --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default (grid = CUs)', d['ms_per_step'])"
 VILA_SFT_ADAMW_GRID=1024 timeout 300 python bench.py --mode sft --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid 1024', d['ms_per_step'])"
done
VILA_SFT_C_ABI=1 timeout 300 python bench.py --mode sft --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C-ABI default', d['ms_per_step'])"
VILA_SFT_C_ABI=1 VILA_SFT_ADAMW_GRID=1024 timeout 300 python bench.py --mode sft --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C-ABI grid 1024', d['ms_per_step'])"
