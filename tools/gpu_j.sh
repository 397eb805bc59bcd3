#!/bin/bash
# round-2 pass J: parity after the prefill last-layer pruning + contraction-major tower backward, SFT A/B, kernel traces, PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_full_depth.py tests/test_gpu_serving.py tests/test_gpu_sampling.py -m gpu -q -x 2>&1 | tail -15 > "$O/r2j_pytest.log"
tail -3 "$O/r2j_pytest.log"
for v in 0 1 0 1; do
  VILA_SFT_CM_VIT=$v timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/r2j_sft_$v.err" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cm_vit=$v ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"
done 2>&1 | tee "$O/r2j_sft_cmvit.log"
# kernel traces
timeout 600 bash tools/profile.sh r2j --no-sft --no-sustain --steps 32 --warmup 8 2>&1 | tail -2
if [ -f "$O/prof_r2j/trace_results.db" ]; then
  python tools/rocpd_summary.py "$O/prof_r2j/trace_results.db" "$O/r2j_bench_kernel_stats.csv"
  python tools/rocpd_timeline.py "$O/prof_r2j/trace_results.db" im2col_kernel argmax_stage2 -1 "$O/r2j_ttft_timeline.txt"
  head -3 "$O/r2j_ttft_timeline.txt"
  rm -f "$O/prof_r2j/trace_results.db"
fi
timeout 600 bash tools/profile.sh r2j_sft --mode sft --steps 3 --warmup 1 2>&1 | tail -2
if [ -f "$O/prof_r2j_sft/trace_results.db" ]; then
  python tools/rocpd_summary.py "$O/prof_r2j_sft/trace_results.db" "$O/r2j_sft_kernel_stats.csv"
  rm -f "$O/prof_r2j_sft/trace_results.db"
fi
# HBM traffic of the decode kernels (two passes) and MFMA utilisation (TTFT + decode, SFT step)
timeout 600 bash tools/pmc.sh r2j --no-sft --no-sustain --steps 8 --warmup 2
for C in FETCH_SIZE WRITE_SIZE; do
  D="$O/pmc_r2j_$C/pmc_results.db"
  if [ -f "$D" ]; then python tools/pmc_summary.py "$D" gemv_kernel > "$O/r2j_pmc_$C.txt" 2>&1; rm -f "$D"; fi
done
timeout 600 bash tools/pmc_mfma.sh r2j_ttft --no-sft --no-sustain --steps 8 --warmup 2 | tail -3
timeout 900 bash tools/pmc_mfma.sh r2j_sft --mode sft --steps 2 --warmup 1 | tail -3
cp "$O/pmc_mfma_r2j_ttft/summary.txt" "$O/r2j_pmc_mfma_ttft_decode.txt" 2>/dev/null
cp "$O/pmc_mfma_r2j_sft/summary.txt" "$O/r2j_pmc_mfma_sft_step.txt" 2>/dev/null
find "$O" -name "*.db" -size +1M -delete
du -sh "$O" | tail -1
