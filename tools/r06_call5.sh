#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/r06_pytest_gpu_mid.log
tail -6 $O/r06_pytest_gpu_mid.log
timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>>$O/r06_call5.err | tail -1 > $O/r06_sft_det.json
python -c "
import json
d=json.loads(open('gpurun_out/r06_sft_det.json').read()); print('sft (deterministic reductions):', d['ms_per_step'], 'ms')"
