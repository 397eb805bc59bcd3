#!/bin/bash
# SECOND GPU call of round 5: the new defaults (K-sliced ring for M < 512, ring PIPE = 2) A/B'd end to end, the finer A/Bs of the round-4 variants
# that the first call left open (W4 LAT, batch-8 norm LAT, SFT EPF), the un-gated tests/test_gpu_run.py with the noise-floor resume test,
# the new kernels of this session (vila_grad_accum_f32, the fixed-point nucleus histograms) and the calibration dump the full-depth fixtures need
# after a change of the prefill's rounding.
#   gpurun --timeout 1500 -- bash tools/r05_second_call.sh          results under gpurun_out/r05_second/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_second; mkdir -p "$O"
line() { python -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2:', d.get('config',{}).get('workload','')[:60], '| value', d['value'], ' ms/step', d['ms_per_step'], ' ttft', d.get('ttft_ms'))
except Exception as e:
    print('$2: FAILED', e); print(open('${1%.json}.err').read()[-600:])"; }
B="--no-sft --no-sustain --no-cpu-baseline"
timeout 900 python -m pytest tests/test_gpu_run.py tests/test_gpu_sampling.py "tests/test_gpu_ops.py::test_grad_accum_f32_is_the_fp32_sum_rounded_once" -m gpu -q -s -x 2>&1 | tail -40 > "$O/pytest_new.log"; tail -6 "$O/pytest_new.log"
timeout 600 python tools/dump_full_depth_hidden.py > "$O/dump_calib.log" 2>&1; tail -4 "$O/dump_calib.log"
# the new defaults end to end: short prompts (S = 289, Lite-3B S = 154) and the headline prompt
for v in 0 1; do
  VILA_RING_SPLITK=$v timeout 300 python bench.py --prompt-tokens 32 $B --steps 32 --warmup 8 > "$O/ring_splitk_$v.json" 2> "$O/ring_splitk_$v.err"; line "$O/ring_splitk_$v.json" "VILA_RING_SPLITK=$v S=289"
  VILA_RING_SPLITK=$v timeout 300 python bench.py --config nvila_lite_3b --prompt-tokens 32 $B --steps 32 --warmup 8 > "$O/ring_splitk_lite_$v.json" 2> "$O/ring_splitk_lite_$v.err"; line "$O/ring_splitk_lite_$v.json" "VILA_RING_SPLITK=$v Lite-3B"
done
for v in 0 2; do
  VILA_RING_PIPE=$v timeout 300 python bench.py $B --steps 32 --warmup 8 > "$O/ring_pipe$v.json" 2> "$O/ring_pipe$v.err"; line "$O/ring_pipe$v.json" "VILA_RING_PIPE=$v S=769"
done
# W4 decode with the batched merge + LAT epilogues + early K/V
for v in 0 1 0 1; do
  VILA_DECODE_LAT=$v timeout 300 python bench.py --w4 $B > "$O/w4_lat$v.json" 2> "$O/w4_lat$v.err"; line "$O/w4_lat$v.json" "W4, VILA_DECODE_LAT=$v"
done
# batched decode (8 rows) with the norm LAT kernel
for v in 0 1; do
  VILA_NORM_LAT=$v timeout 300 python bench.py --batch 8 $B --steps 48 --warmup 8 > "$O/batch8_normlat$v.json" 2> "$O/batch8_normlat$v.err"; line "$O/batch8_normlat$v.json" "batch 8, VILA_NORM_LAT=$v"
done
# the SFT step with the 256x256 kernel's epilogue prefetch
for v in 0 1 0 1; do
  VILA_GEMM256_EPF=$v timeout 400 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/sft_epf$v.err" | tee "$O/sft_epf$v.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VILA_GEMM256_EPF=$v: sft ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"
done
