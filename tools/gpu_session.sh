#!/bin/bash
# One GPU session of round 3: parity first, then the A/B measurements of this round's kernels.  Run from the repo root on the GPU box:
#   gpurun -- bash tools/gpu_session.sh <tag> [steps...]       results under gpurun_out/<tag>/
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-s}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$O"
for step in "$@"; do
case $step in
  gemm_bwd)  timeout 300 tools/gemm_bench bwd > "$O/gemm_bench_bwd.log" 2>&1; tail -40 "$O/gemm_bench_bwd.log" ;;
  gemm_fwd)  timeout 300 tools/gemm_bench fwd > "$O/gemm_bench_fwd.log" 2>&1; tail -30 "$O/gemm_bench_fwd.log" ;;
  tests)     timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > "$O/pytest.log"; tail -30 "$O/pytest.log" ;;
  tests_all) timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 > "$O/pytest_all.log"; tail -40 "$O/pytest_all.log" ;;
  tests_rest) timeout 1500 python -m pytest tests/test_gpu_serving.py tests/test_gpu_train.py tests/test_gpu_w4.py tests/test_video_encoders.py tests/test_w8a8.py -m gpu -q 2>&1 | tail -40 > "$O/pytest_rest.log"; tail -40 "$O/pytest_rest.log" ;;
  attn_bwd)  VILA_ATTN_BWD=v1 timeout 300 python tools/microbench.py attn_bwd > "$O/attn_bwd_v1.log" 2>&1; timeout 300 python tools/microbench.py attn_bwd > "$O/attn_bwd_new.log" 2>&1
             paste -d'\n' "$O/attn_bwd_v1.log" "$O/attn_bwd_new.log" ;;
  decode_warm) for mb in 0 64 128 192 270; do VILA_DECODE_WARM_MB=$mb timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline --steps 64 --warmup 8 > "$O/warm_$mb.json" 2> "$O/warm_$mb.err"; python -c "
import json; d=json.loads(open('$O/warm_$mb.json').read().strip().splitlines()[-1]); print('warm MB $mb: value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 "$O/warm_$mb.err"; done ;;
  seam)      timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -k "seam or per_bucket" 2>&1 | tail -8 ;;
  ex)        timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -k "leftover or tail_rows or gateup or splitk or gemm_plain" 2>&1 | tail -8
             for e in 0 1; do VILA_GEMM_EX=$e timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline --steps 32 --warmup 8 > "$O/ex_$e.json" 2> "$O/ex_$e.err"; python -c "
import json; d=json.loads(open('$O/ex_$e.json').read().strip().splitlines()[-1]); print('EX=$e: ttft', d['ttft_ms'], 'value', d['value'])" || tail -3 "$O/ex_$e.err"; done
             for e in 0 1; do VILA_GEMM_EX=$e timeout 300 tools/gemm_bench fwd > "$O/gemm_fwd_ex$e.log" 2>&1; grep -E "sched=0 " "$O/gemm_fwd_ex$e.log" | head -24; done ;;
  batch)     timeout 900 python -m pytest tests/test_gpu_batch_decode.py -m gpu -q 2>&1 | tail -25
             for b in 2 4 8 16; do timeout 300 python bench.py --batch $b --steps 48 --warmup 8 > "$O/batch_$b.json" 2> "$O/batch_$b.err"; python -c "
import json; d=json.loads(open('$O/batch_$b.json').read().strip().splitlines()[-1]); print('batch $b: aggregate', d['value'], 'tok/s  ms/step', d['ms_per_step'], 'hbm frac', d['roofline']['frac'])" || tail -3 "$O/batch_$b.err"; done ;;
  prof_batch) timeout 600 bash tools/profile.sh batch8 --batch 8 --steps 32 --warmup 4 | tail -2
             python tools/rocpd_summary.py "$GRAFT_REPO_ROOT/gpurun_out/prof_batch8/trace_results.db" "$O/batch8_kernel_stats.csv"; head -14 "$O/batch8_kernel_stats.csv"; rm -f "$GRAFT_REPO_ROOT/gpurun_out/prof_batch8/trace_results.db" ;;
  tower)     cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$O/prof_tower" -o trace -- python "$GRAFT_REPO_ROOT/tools/microbench.py" tower 2>&1 | grep -i "tower" ; cd "$GRAFT_REPO_ROOT"
             python tools/rocpd_summary.py "$O/prof_tower/trace_results.db" "$O/tower_kernel_stats.csv"; head -16 "$O/tower_kernel_stats.csv"; rm -f "$O/prof_tower/trace_results.db" ;;
  batch_test) timeout 900 python -m pytest tests/test_gpu_batch_decode.py -m gpu -q 2>&1 | tail -8 ;;
  tests_new) timeout 1200 python -m pytest tests/test_gpu_baseline_configs.py tests/test_dynamic_s2.py tests/test_gpu_sampling.py -m gpu -q 2>&1 | tail -30 > "$O/pytest_new.log"; tail -30 "$O/pytest_new.log" ;;
  tests_ops) timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -m gpu -q 2>&1 | tail -30 > "$O/pytest_ops.log"; tail -30 "$O/pytest_ops.log" ;;
  attn)      VILA_ATTN_FWD=v1 timeout 300 python tools/microbench.py attn > "$O/attn_v1.log" 2>&1; timeout 300 python tools/microbench.py attn > "$O/attn_new.log" 2>&1
             paste -d'\n' "$O/attn_v1.log" "$O/attn_new.log" ;;
  sft)       for i in 1 2; do timeout 400 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/sft_$i.err" | tee "$O/sft_$i.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sft ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"; done ;;
  sft_s2)    timeout 600 python bench.py --mode sft --dynamic-s2 --micro-batch 1 --steps 3 --warmup 1 2>"$O/sft_s2.err" | tee "$O/sft_s2.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sft dynamic_s2 ->', d['ms_per_step'], 'ms  frac', d['roofline']['frac'])" ;;
  bench)     timeout 600 python bench.py > "$O/bench.json" 2> "$O/bench.err"; python - "$O/bench.json" <<'P'
import json,sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], "ttft", d["ttft_ms"], "prefill frac", d["prefill"]["roofline"]["frac"], "sft", d.get("sft", {}).get("ms_per_step"), "roofline", d["roofline"]["frac"])
except Exception as e:
    print("bench parse failed", e)
P
             ;;
  bench_fast) timeout 400 python bench.py --no-sft --no-sustain --no-cpu-baseline > "$O/bench_fast.json" 2> "$O/bench_fast.err"; python -c "
import json; d=json.loads(open('$O/bench_fast.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ttft', d['ttft_ms'], 'prefill frac', d['prefill']['roofline']['frac'])" ;;
  video)     for args in "--mode video" "--mode video --tsp"; do timeout 400 python bench.py $args 2>>"$O/video.err" | tail -1 | tee -a "$O/video.jsonl" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], d['value'], 'ms encode', d['encode_ms'], 'prefill', d['llm_prefill_ms'])"; done ;;
  chain_tests) timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_batch_decode.py tests/test_gpu_serving.py tests/test_gpu_sampling.py -m gpu -q -x 2>&1 | tail -25 > "$O/pytest_chain.log"; tail -25 "$O/pytest_chain.log" ;;
  full_depth) timeout 1200 python -m pytest tests/test_gpu_full_depth.py -m gpu -q -s -k "${FD_K:-full_depth}" 2>&1 | grep -v "^$" > "$O/pytest_full_depth.log"; grep -n "full depth\|forward loss\|Error\|passed\|failed" "$O/pytest_full_depth.log" | cut -c1-600 | tail -30 ;;
  chain_ab)  for cv in "0 95" "1 95" "1 80" "1 110"; do set -- $cv; VILA_DECODE_CHAIN=$1 VILA_DECODE_CHAIN_PRED=$2 timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline --steps 64 --warmup 8 > "$O/chain_$1_$2.json" 2> "$O/chain_$1_$2.err"; python -c "
import json; d=json.loads(open('$O/chain_$1_$2.json').read().strip().splitlines()[-1]); print('chain=$1 pred=$2: value', d['value'], 'ms/step', d['ms_per_step'], 'ttft', d['ttft_ms'], 'gateup us', d['roofline']['avg_launch_us'])" || tail -5 "$O/chain_$1_$2.err"; done ;;
  chain_prof) timeout 600 bash tools/profile.sh chain --no-sft --no-sustain --no-cpu-baseline --steps 32 --warmup 8 2>&1 | tail -2
             python tools/rocpd_summary.py "$GRAFT_REPO_ROOT/gpurun_out/prof_chain/trace_results.db" "$O/chain_kernel_stats.csv"; head -8 "$O/chain_kernel_stats.csv"
             python tools/rocpd_rows.py "$GRAFT_REPO_ROOT/gpurun_out/prof_chain/trace_results.db" decode_prologue_kernel decode_advance_kernel -3 "$O/chain_token_rows.txt"; head -34 "$O/chain_token_rows.txt"
             rm -f "$GRAFT_REPO_ROOT/gpurun_out/prof_chain/trace_results.db" ;;
  bpc2)      VILA_DECODE_CHAIN=0 VILA_GEMV_BPC=2 timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline --steps 64 --warmup 8 > "$O/bpc2.json" 2> "$O/bpc2.err"; python -c "
import json; d=json.loads(open('$O/bpc2.json').read().strip().splitlines()[-1]); print('chain=0 bpc=2: value', d['value'], 'ms/step', d['ms_per_step'], 'gateup us', d['roofline']['avg_launch_us'])" || tail -5 "$O/bpc2.err" ;;
  fuse_ab)   for c in 0 1 0 1; do VILA_FUSE_NORM=$c timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline --steps 32 --warmup 8 > "$O/fuse_$c.json" 2> "$O/fuse_$c.err"; python -c "
import json; d=json.loads(open('$O/fuse_$c.json').read().strip().splitlines()[-1]); print('fuse_norm=$c: ttft', d['ttft_ms'], 'encode', d['prefill']['encode_images_ms'], 'value', d['value'])" || tail -5 "$O/fuse_$c.err"; done ;;
  fuse_tests) timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_size.py tests/test_gpu_baseline_configs.py tests/test_video_encoders.py tests/test_dynamic_s2.py -m gpu -q -x 2>&1 | tail -8 ;;
  smoke)     python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
  *) echo "unknown step $step" ;;
esac
done
