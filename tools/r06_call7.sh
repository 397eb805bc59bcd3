#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -k "nccl" 2>&1 | tail -4
for a in direct all_reduce; do
VILA_BENCH_FORCE_DIST=1 VILA_GRAD_EXCHANGE=$a timeout 400 python bench.py --mode sft --steps 3 --warmup 1 2>$O/r06_sft_forcedist_$a.err | tail -1 > $O/r06_sft_forcedist_$a.json
python -c "
import json
d=json.loads(open('$O/r06_sft_forcedist_$a.json').read()); c=d['config']; print('$a', d['ms_per_step'], c.get('exchange_algo'), c.get('exchange_bytes'), c.get('rccl_ranks_seen'), c.get('rccl_backend'))"
done
