#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['ms_per_step'], 'ms loss', d['loss'])" | tee -a $O/r06_adamw_stream_ab.log; }
for i in 1 2; do
run VILA_SFT_ADAMW_STREAM=0
run VILA_SFT_ADAMW_STREAM=2
run VILA_SFT_ADAMW_STREAM=3
run VILA_SFT_ADAMW_STREAM=4
run VILA_SFT_ADAMW_STREAM=3 VILA_SFT_ADAMW_LDS=0
run VILA_SFT_ADAMW_STREAM=2 VILA_SFT_ADAMW_LDS=0
done
