#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --mode sft --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['ms_per_step'], 'ms loss', d['loss'])" | tee -a $O/r06_adamw_grid_ab.log; }
for i in 1 2; do
for g in 128 192 224 256 288 320 1024; do run VILA_SFT_ADAMW_GRID=$g; done
done
