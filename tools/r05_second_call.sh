#!/bin/bash
# SECOND GPU call of round 5 (after tools/r05_first_call.sh): the finer A/Bs of the unmeasured round-4 variants.
#   gpurun --timeout 1500 -- bash tools/r05_second_call.sh          results under gpurun_out/r05_second/   (~20 min)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_second; mkdir -p "$O"
line() { python -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2:', d.get('config',{}).get('workload','')[:60], '| value', d['value'], ' ms/step', d['ms_per_step'], ' ttft', d.get('ttft_ms'))
except Exception as e:
    print('$2: FAILED', e); print(open('${1%.json}.err').read()[-600:])"; }
B="--no-sft --no-sustain --no-cpu-baseline"
# each decode-latency variant alone
for v in VILA_GEMV_GAIN_EARLY VILA_GEMV_MERGE_BATCH VILA_DECODE_ATTN_EARLY_KV VILA_GEMV_X_FIRST; do
  env $v=1 timeout 300 python bench.py $B > "$O/decode_$v.json" 2> "$O/decode_$v.err"; line "$O/decode_$v.json" "$v=1 alone"
done
# W4 decode with the batched merge + LAT epilogues + early K/V
for v in 0 1 0 1; do
  VILA_DECODE_LAT=$v timeout 300 python bench.py --w4 $B > "$O/w4_lat$v.json" 2> "$O/w4_lat$v.err"; line "$O/w4_lat$v.json" "W4, VILA_DECODE_LAT=$v"
done
# ring tile switch at S = 769, K-sliced ring on the short prompts
for v in 12 16; do
  VILA_RING_BIG=$v timeout 300 python bench.py $B --steps 32 --warmup 8 > "$O/ring_big_$v.json" 2> "$O/ring_big_$v.err"; line "$O/ring_big_$v.json" "VILA_RING_BIG=$v"
done
for v in 0 1; do
  VILA_RING_SPLITK=$v timeout 300 python bench.py --prompt-tokens 32 $B --steps 32 --warmup 8 > "$O/ring_splitk_$v.json" 2> "$O/ring_splitk_$v.err"; line "$O/ring_splitk_$v.json" "VILA_RING_SPLITK=$v S=289"
  VILA_RING_SPLITK=$v timeout 300 python bench.py --config nvila_lite_3b --prompt-tokens 32 $B --steps 32 --warmup 8 > "$O/ring_splitk_lite_$v.json" 2> "$O/ring_splitk_lite_$v.err"; line "$O/ring_splitk_lite_$v.json" "VILA_RING_SPLITK=$v Lite-3B"
done
# batched decode (8 rows) with the norm LAT kernel
for v in 0 1; do
  VILA_NORM_LAT=$v timeout 300 python bench.py --batch 8 $B --steps 48 --warmup 8 > "$O/batch8_normlat$v.json" 2> "$O/batch8_normlat$v.err"; line "$O/batch8_normlat$v.json" "batch 8, VILA_NORM_LAT=$v"
done
# the SFT step with the 256x256 kernel's epilogue prefetch (and the ring PIPE for the tower's shapes)
for v in 0 1; do
  VILA_GEMM256_EPF=$v VILA_RING_PIPE=$v timeout 400 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/sft_epf$v.err" | tee "$O/sft_epf$v.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VILA_GEMM256_EPF=$v VILA_RING_PIPE=$v: sft ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"
done
