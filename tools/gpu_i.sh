#!/bin/bash
# round-2 validation pass: GEMM harness (default schedules vs round-1 schedule), full GPU test-suite, default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 120 tools/gemm_bench all > gpurun_out/r2i_gemm_bench.log 2>&1
grep -c MISMATCH gpurun_out/r2i_gemm_bench.log
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r2i_pytest.log
tail -5 gpurun_out/r2i_pytest.log
timeout 600 python bench.py > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r2i_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ttft", d["ttft_ms"], "prefill frac", d["prefill"]["roofline"]["frac"], "sft", d.get("sft", {}).get("ms_per_step"), "sustained", d.get("sustained", {}).get("tokens_per_s"))
except Exception as e:
    print("bench parse failed", e)
P
tail -3 gpurun_out/r2i_bench.err
