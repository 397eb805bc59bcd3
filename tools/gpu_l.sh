#!/bin/bash
# round-2 pass L: full GPU suite on the new default schedules / launch policy, bench lines, TTFT timeline
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > "$O/r2l_pytest.log"
tail -3 "$O/r2l_pytest.log"
for i in 1 2; do
  timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/r2l_sft_$i.err" | tee "$O/r2l_sft_$i.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sft ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"
done
VILA_SFT_C_ABI=1 timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/r2l_sft_c.err" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sft c-abi ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"
timeout 600 python bench.py > "$O/r2l_bench.json" 2> "$O/r2l_bench.err"
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r2l_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ttft", d["ttft_ms"], "prefill frac", d["prefill"]["roofline"]["frac"], "sft", d.get("sft", {}).get("ms_per_step"), "sustained", d.get("sustained", {}).get("tokens_per_s"))
except Exception as e:
    print("bench parse failed", e)
P
timeout 600 bash tools/profile.sh r2l --no-sft --no-sustain --steps 32 --warmup 8 2>&1 | tail -1
if [ -f "$O/prof_r2l/trace_results.db" ]; then
  python tools/rocpd_summary.py "$O/prof_r2l/trace_results.db" "$O/r2l_bench_kernel_stats.csv"
  python tools/rocpd_timeline.py "$O/prof_r2l/trace_results.db" im2col_kernel argmax_stage2 -4 "$O/r2l_ttft_timeline.txt"
  head -3 "$O/r2l_ttft_timeline.txt"
  rm -f "$O/prof_r2l/trace_results.db"
fi
find "$O" -name "*.db" -size +1M -delete
