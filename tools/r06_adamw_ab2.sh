#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['ms_per_step'], 'ms loss', d['loss'])" | tee -a $O/r06_adamw_grid_ab.log; }
for i in 1 2; do
run VILA_SFT_ADAMW_GRID=1024
run VILA_SFT_ADAMW_GRID=768
run VILA_SFT_ADAMW_GRID=512
run VILA_SFT_ADAMW_GRID=384
run VILA_SFT_ADAMW_GRID=256
run VILA_SFT_ADAMW_GRID=2048
done
