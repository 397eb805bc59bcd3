#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "persistent" 2>&1 | tail -8
python tools/decode_persist_trace.py --out $O/r06_persist_trace_$1.txt 2>&1 | tail -12
for m in 0 1; do
  VILA_DECODE_PERSIST=$m timeout 300 python bench.py --no-sft --no-cpu-baseline --no-sustain 2>>$O/r06_call3.err | tail -1 > $O/r06_decode_persist${m}_$1.json
  python -c "
import json
d=json.loads(open('$O/r06_decode_persist${m}_$1.json').read()); print('persist=$m: value', d['value'], 'ms/step', d['ms_per_step'], 'ttft', d.get('ttft_ms'))"
done
