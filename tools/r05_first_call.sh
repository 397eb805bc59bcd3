#!/bin/bash
# FIRST GPU call of round 5: everything that was written at the end of round 4 WITHOUT a GPU (0.6 GPU-minutes were left) gets its first run.
#   gpurun --timeout 1500 -- bash tools/r05_first_call.sh          results under gpurun_out/r05_first/
# 1. the gated tests (tests/test_gpu_run.py: accumulated update, resumed run, the 3- / 4-stage 128x128 ring, the K-sliced ring) — apart from the suite
# 2. tools/gemm_bench prering: the new ring variants (force_tile 9 / 10) vs the current choices on the prefill / tower shapes, cold weights
# 3. TTFT A/B of VILA_RING_PIPE = 1 (the ring kernels' fragment schedule) and of the dispatch switch VILA_RING_BIG = 0 / 12 / 16 on the default bench line (no SFT, no sustained loop, no CPU leg)
# 3a. decode: each latency variant alone (gain early, merge batch, early K/V, x first), then VILA_DECODE_LAT = 0 / 1 twice; W4 with the batched merge
# 3b. VILA_RING_SPLITK = 0 / 1 on the short-prompt lines (gemm_ring_splitk.hip)
# 4. the suite itself (the round-4 late commits after the last full run: chat template, prepare_tokenizer, stop_token_ids)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_first; mkdir -p "$O"
VILA_TEST_UNVERIFIED=1 timeout 600 python -m pytest tests/test_gpu_run.py -m gpu -q -s 2>&1 | tail -40 > "$O/pytest_unverified.log"; tail -5 "$O/pytest_unverified.log"
[ -x tools/gemm_bench ] || hipcc -O2 -std=c++17 tools/gemm_bench.cpp -o tools/gemm_bench -Iinclude -Lvila_amd/lib -lvila_hip -Wl,-rpath,'$ORIGIN/../vila_amd/lib'
timeout 300 tools/gemm_bench prering > "$O/gemm_bench_prering.log" 2>&1; tail -72 "$O/gemm_bench_prering.log"
for v in 1 2; do
  VILA_RING_PIPE=$v timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline --steps 32 --warmup 8 > "$O/ring_pipe$v.json" 2> "$O/ring_pipe$v.err"
  python -c "
import json; d=json.loads(open('$O/ring_pipe$v.json').read().strip().splitlines()[-1]); print('VILA_RING_PIPE=$v: ttft', d['ttft_ms'], 'ms  decode', d['value'], 'tok/s')" || tail -3 "$O/ring_pipe$v.err"
done
# decode: the RMSNorm gain by LDS-DMA ahead of x (stage_x_ge) — three runs each, the headline metric
for v in VILA_GEMV_GAIN_EARLY VILA_GEMV_MERGE_BATCH VILA_DECODE_ATTN_EARLY_KV VILA_GEMV_X_FIRST; do
  env $v=1 timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline > "$O/decode_$v.json" 2> "$O/decode_$v.err"
  python -c "
import json; d=json.loads(open('$O/decode_$v.json').read().strip().splitlines()[-1]); print('$v=1 alone: decode', d['value'], 'tok/s  ms/step', d['ms_per_step'], ' dominant-kernel frac', d['roofline']['frac'])" || tail -3 "$O/decode_$v.err"
done
for v in 0 1 0 1; do
  VILA_DECODE_LAT=$v timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline > "$O/decode_lat$v.json" 2> "$O/decode_lat$v.err"
  python -c "
import json; d=json.loads(open('$O/decode_lat$v.json').read().strip().splitlines()[-1]); print('VILA_DECODE_LAT=$v (gain early + merge batch + early K/V): decode', d['value'], 'tok/s  ms/step', d['ms_per_step'], ' dominant-kernel frac', d['roofline']['frac'])" || tail -3 "$O/decode_lat$v.err"
done
for v in 0 1 0 1; do
  VILA_DECODE_LAT=$v timeout 300 python bench.py --w4 --no-sft --no-sustain --no-cpu-baseline > "$O/w4_mb$v.json" 2> "$O/w4_mb$v.err"
  python -c "
import json; d=json.loads(open('$O/w4_mb$v.json').read().strip().splitlines()[-1]); print('W4 decode, VILA_DECODE_LAT=$v (batched merge + LAT epilogues + early K/V):', d['value'], 'tok/s  ms/step', d['ms_per_step'])" || tail -3 "$O/w4_mb$v.err"
done
for v in 0 12 16; do
  VILA_RING_BIG=$v timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline --steps 32 --warmup 8 > "$O/ring_big_$v.json" 2> "$O/ring_big_$v.err"
  python -c "
import json; d=json.loads(open('$O/ring_big_$v.json').read().strip().splitlines()[-1]); print('VILA_RING_BIG=$v: ttft', d['ttft_ms'], 'ms  decode', d['value'], 'tok/s')" || tail -3 "$O/ring_big_$v.err"
done
# 3b. the K-sliced ring on SHORT prompts (VILA_RING_SPLITK = 0 / 1): configs[1]'s 32-token prompt (S = 289) and the Lite-3B line
for v in 0 1; do
  VILA_RING_SPLITK=$v timeout 300 python bench.py --prompt-tokens 32 --no-sft --no-sustain --no-cpu-baseline --steps 32 --warmup 8 > "$O/ring_splitk_$v.json" 2> "$O/ring_splitk_$v.err"
  VILA_RING_SPLITK=$v timeout 300 python bench.py --config nvila_lite_3b --prompt-tokens 32 --no-sft --no-sustain --no-cpu-baseline --steps 32 --warmup 8 > "$O/ring_splitk_lite_$v.json" 2> "$O/ring_splitk_lite_$v.err"
  python -c "
import json
for f in ('$O/ring_splitk_$v.json', '$O/ring_splitk_lite_$v.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print('VILA_RING_SPLITK=$v', d['config'].get('workload'), ': ttft', d['ttft_ms'], 'ms  decode', d['value'], 'tok/s')" || tail -3 "$O/ring_splitk_$v.err"
done
# 3c. the SFT step with the 256x256 kernel's epilogue prefetch (VILA_GEMM256_EPF = 0 / 1)
for v in 0 1; do
  VILA_GEMM256_EPF=$v timeout 400 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/sft_epf$v.err" | tee "$O/sft_epf$v.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VILA_GEMM256_EPF=$v: sft ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"
done
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -25 > "$O/pytest.log"; tail -3 "$O/pytest.log"
