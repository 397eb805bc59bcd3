#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_batch_decode.py tests/test_gpu_full_depth.py -m gpu -q -x -k "not backward_probe and not sft_forward" 2>&1 | grep "passed\|failed" | tail -3
for i in 1 2; do
for cfgv in "0 0" "1 0" "1 1" "1 2"; do
  set -- $cfgv
  VILA_GEMV_CU_MAP=$1 VILA_GEMV_SKEW=$2 timeout 300 python bench.py --no-sft --no-cpu-baseline --no-sustain 2>>$O/r06_call9.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cu_map=$1 skew=$2: value', d['value'], 'ms/step', d['ms_per_step'], 'dominant kernel us', d['roofline']['avg_launch_us'])" | tee -a $O/r06_gemv_cu_map_ab.log
done
done
