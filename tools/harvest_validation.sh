#!/bin/bash
# Copy the summaries tools/validate_gpu.sh left under gpurun_out/ into profiles/ under this round's names.  usage: bash tools/harvest_validation.sh r04 [suffix]
R=${1:-r04}; SFX=${2:-}
O=gpurun_out; P=profiles
cpy() { [ -s "$O/$1" ] && cp "$O/$1" "$P/${R}_$2${SFX}" && echo "  $P/${R}_$2${SFX}"; }
cpy val_bench.json bench_final.json
cpy val_pytest.log pytest_gpu_final.log
cpy val_other_modes.jsonl bench_other_modes.jsonl
cpy val_batch_decode.jsonl batch_decode.jsonl
cpy val_bench_lite3b.json bench_lite3b.json
cpy val_pmc_hbm_counters.txt pmc_hbm_counters.txt
cpy val_pmc_traffic.json pmc_traffic.json
cpy val_pmc_gemm_sft.txt pmc_gemm_sft.txt
cpy val_pmc_gemm_sft.json pmc_gemm_sft.json
cpy val_pmc_mfma_sft_step.txt pmc_mfma_sft_step.txt
cpy val_pmc_mfma_ttft_decode.txt pmc_mfma_ttft_decode.txt
cpy val_sft_kernel_stats.csv sft_step_kernel_stats.csv
cpy val_bench_kernel_stats.csv bench_kernel_stats.csv
cpy val_ttft_timeline.txt ttft_timeline.txt
cpy val_sft_forcedist.json sft_forcedist_world1_nccl.json
cpy val_bench_forcedist.json bench_forcedist_world1_nccl.json
cpy val_persist_ab.log decode_persist_ab_final.log
: > "$P/${R}_sft_step_final${SFX}.jsonl"
for f in val_sft_1.json val_sft_2.json val_sft_c.json val_sft_s2.json; do [ -s "$O/$f" ] && tail -1 "$O/$f" >> "$P/${R}_sft_step_final${SFX}.jsonl"; done
echo "  $P/${R}_sft_step_final${SFX}.jsonl"
