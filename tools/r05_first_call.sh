#!/bin/bash
# FIRST GPU call of round 5: everything written at the end of round 4 WITHOUT a GPU (0.6 GPU-minutes were left) gets its first run.
#   gpurun --timeout 1700 -- bash tools/r05_first_call.sh          results under gpurun_out/r05_first/   (~25 min)
# 1. the gated tests (tests/test_gpu_run.py), apart from the suite so that they cannot stop it
# 2. tools/gemm_bench prering: every new ring variant (force_tile 9..19) with its error column, cold weights
# 3. the headline A/Bs: decode with VILA_DECODE_LAT = 0 / 1 (twice), TTFT with VILA_RING_PIPE = 0 / 1 / 2
# 4. the suite itself (the round-4 late commits after the last full run: chat template, prepare_tokenizer, stop_token_ids)
# tools/r05_second_call.sh holds the finer A/Bs (each decode variant alone, W4, ring tile / K-slicing switches, the SFT step with EPF).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_first; mkdir -p "$O"
line() { python -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2:', 'decode', d['value'], 'tok/s  ms/step', d['ms_per_step'], ' ttft', d.get('ttft_ms'), 'ms  dominant-kernel frac', (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print('$2: FAILED', e); print(open('${1%.json}.err').read()[-600:])"; }
VILA_TEST_UNVERIFIED=1 timeout 900 python -m pytest tests/test_gpu_run.py -m gpu -q -s 2>&1 | tail -60 > "$O/pytest_unverified.log"; tail -8 "$O/pytest_unverified.log"
[ -x tools/gemm_bench ] || hipcc -O2 -std=c++17 tools/gemm_bench.cpp -o tools/gemm_bench -Iinclude -Lvila_amd/lib -lvila_hip -Wl,-rpath,'$ORIGIN/../vila_amd/lib'
timeout 400 tools/gemm_bench prering > "$O/gemm_bench_prering.log" 2>&1; grep -c MISMATCH "$O/gemm_bench_prering.log"; tail -60 "$O/gemm_bench_prering.log"
for v in 0 1 0 1; do
  VILA_DECODE_LAT=$v timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline > "$O/decode_lat$v.json" 2> "$O/decode_lat$v.err"; line "$O/decode_lat$v.json" "VILA_DECODE_LAT=$v"
done
for v in 0 1 2; do
  VILA_RING_PIPE=$v timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline --steps 32 --warmup 8 > "$O/ring_pipe$v.json" 2> "$O/ring_pipe$v.err"; line "$O/ring_pipe$v.json" "VILA_RING_PIPE=$v"
done
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -25 > "$O/pytest.log"; tail -3 "$O/pytest.log"
