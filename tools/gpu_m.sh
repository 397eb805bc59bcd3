#!/bin/bash
# round-2 pass M: new GEMM tests, W8A8 bench line, MFMA counters of the SFT step and of TTFT + decode on the final defaults
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "schedule or whole_rounds or gemm_t" 2>&1 | tail -8 > "$O/r2m_pytest_gemm.log"
tail -3 "$O/r2m_pytest_gemm.log"
timeout 300 python bench.py --w8-vit --no-sft --no-sustain --no-cpu-baseline > "$O/r2m_bench_w8_vit.json" 2> "$O/r2m_bench_w8_vit.err"
python -c "
import json
d=json.loads(open('gpurun_out/r2m_bench_w8_vit.json').read().strip().splitlines()[-1]); print('w8 vit: ttft', d['ttft_ms'], 'encode', d['prefill']['encode_images_ms'], 'value', d['value'])"
timeout 900 bash tools/pmc_mfma.sh r2m_sft --mode sft --steps 2 --warmup 1 | tail -2
cp "$O/pmc_mfma_r2m_sft/summary.txt" "$O/r2m_pmc_mfma_sft_step.txt" 2>/dev/null
timeout 600 bash tools/pmc_mfma.sh r2m_ttft --no-sft --no-sustain --steps 8 --warmup 2 | tail -2
cp "$O/pmc_mfma_r2m_ttft/summary.txt" "$O/r2m_pmc_mfma_ttft_decode.txt" 2>/dev/null
timeout 600 bash tools/profile.sh r2m_sft --mode sft --steps 3 --warmup 1 2>&1 | tail -1
if [ -f "$O/prof_r2m_sft/trace_results.db" ]; then
  python tools/rocpd_summary.py "$O/prof_r2m_sft/trace_results.db" "$O/r2m_sft_kernel_stats.csv"
  rm -f "$O/prof_r2m_sft/trace_results.db"
fi
find "$O" -name "*.db" -size +1M -delete
head -12 "$O/r2m_pmc_mfma_sft_step.txt"
