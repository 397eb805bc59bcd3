#!/bin/bash
# round 6, first GPU call: the gradient cosines of every SFT parity test at the stated bound (dump), the AdamW placement A/B, the baseline bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
VILA_DUMP_COS=$O/r06_grad_cos.jsonl timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q 2>&1 | tail -60 > $O/r06_call1_train_tests.log
tail -5 $O/r06_call1_train_tests.log
for i in 1 2; do
  timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>>$O/r06_call1.err | tail -1 > $O/r06_sft_base_$i.json
  VILA_SFT_OPT_STREAM=0 timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>>$O/r06_call1.err | tail -1 > $O/r06_sft_optdefer_$i.json
done
python - <<'P'
import json
for n in ("sft_base_1","sft_optdefer_1","sft_base_2","sft_optdefer_2"):
    try:
        d=json.loads(open(f"gpurun_out/r06_{n}.json").read()); print(n, d["ms_per_step"], "ms")
    except Exception as e: print(n, "failed", e)
P
timeout 600 python bench.py > $O/r06_bench_base.json 2>> $O/r06_call1.err
python -c "
import json
d=json.loads(open('gpurun_out/r06_bench_base.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ttft', d.get('ttft_ms'), 'sft', (d.get('sft') or {}).get('ms_per_step'))"
