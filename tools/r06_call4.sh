#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "persistent" 2>&1 | tail -3
for sk in 0 1 2 3; do
  echo "== skew $sk"
  VILA_DECODE_PERSIST_SKEW=$sk python tools/decode_persist_trace.py --out $O/r06_persist_trace_skew$sk.txt 2>&1 | grep "persistent layers\|gate/up\|barrier:\|block % 8"
done
