#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
 (cd _old && VILA_SFT_C_ABI=1 timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old C-ABI', d['ms_per_step'], d['loss'])")
 VILA_SFT_C_ABI=1 timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new C-ABI', d['ms_per_step'], d['loss'])"
done
