#!/bin/bash
# VAL_LIGHT=1: the test-suite, one SFT line, the default bench line, the other modes, smoke and the two kernel traces only (no PMC passes,
# no batch / Lite-3B / C-ABI / forced-group lines) — for a late re-validation when only a few GPU-minutes are left.
# Full GPU validation pass: test-suite, every bench line (default, SFT, C-ABI SFT, forced process group, other modes), MFMA counters,
# kernel traces and the TTFT timeline.  Run on the GPU box from the repo root (gpurun -- bash tools/validate_gpu.sh); results under gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p "$O"
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -25 > "$O/val_pytest.log"
tail -3 "$O/val_pytest.log"
for i in 1 $([ -z "$VAL_LIGHT" ] && echo 2); do
  timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/val_sft_$i.err" | tee "$O/val_sft_$i.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sft ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"
done
if [ -z "$VAL_LIGHT" ]; then
VILA_SFT_C_ABI=1 timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>"$O/val_sft_c.err" | tee "$O/val_sft_c.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sft c-abi ->', d['ms_per_step'], 'ms  loss', d.get('loss'))"
VILA_BENCH_FORCE_DIST=1 timeout 300 python bench.py --mode sft --steps 3 --warmup 1 2>"$O/val_sft_forcedist.err" | tee "$O/val_sft_forcedist.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('sft force-dist (nccl, world 1) ->', d['ms_per_step'], 'ms')"
# the default line with the process group forced up in a world of one: the SFT side measurement takes the RCCL path the driver's --gpus N takes
VILA_BENCH_FORCE_DIST=1 timeout 400 python bench.py --no-sustain --no-cpu-baseline --steps 16 --warmup 4 2>"$O/val_bench_forcedist.err" | tail -1 > "$O/val_bench_forcedist.json"
python -c "
import json
d = json.loads(open('gpurun_out/val_bench_forcedist.json').read()); s = d.get('sft') or {}
print('bench (forced nccl group, world 1): value', d['value'], '| sft', s.get('ms_per_step'), 'ms, exchange_active', s.get('exchange_active'), 'bytes', s.get('exchange_bytes'), s.get('error'))
"
fi
timeout 600 python bench.py > "$O/val_bench.json" 2> "$O/val_bench.err"
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/val_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ttft", d["ttft_ms"], "prefill frac", d["prefill"]["roofline"]["frac"], "sft", d.get("sft", {}).get("ms_per_step"), "sustained", d.get("sustained", {}).get("tokens_per_s"))
except Exception as e:
    print("bench parse failed", e)
P
: > "$O/val_other_modes.jsonl"
# the persistent decode layers kernel (opt-in) beside the default per-kernel step, same box
for m in 0 1 0 1; do
  VILA_DECODE_PERSIST=$m timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline 2>>"$O/val_persist.err" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('VILA_DECODE_PERSIST=$m: value', d['value'], 'tok/s, ms/step', d['ms_per_step'], '|', d.get('config', {}).get('decode'))" | tee -a "$O/val_persist_ab.log"
done
for args in "--w4" "--w8-vit" "--dynamic-s2" "--mode video" "--mode video --tsp" "--prompt-tokens 32"; do
  timeout 400 python bench.py $args --no-sft --no-sustain --no-cpu-baseline 2>>"$O/val_other_modes.err" | tail -1 >> "$O/val_other_modes.jsonl"
done
python - <<'P'
import json
for line in open("gpurun_out/val_other_modes.jsonl"):
    try:
        d = json.loads(line); print(d.get("metric", "")[:60], "|", d.get("value"), d.get("unit"), "| ttft", d.get("ttft_ms"), "| ms/step", d.get("ms_per_step"))
    except Exception as e:
        print("bad line", e)
P
if [ -z "$VAL_LIGHT" ]; then
: > "$O/val_batch_decode.jsonl"
for b in 2 4 8 16; do
  timeout 300 python bench.py --batch $b --steps 32 --warmup 4 2>>"$O/val_batch_decode.err" | tail -1 >> "$O/val_batch_decode.jsonl"
done
python -c "
import json
for line in open('gpurun_out/val_batch_decode.jsonl'):
    d = json.loads(line); print('batch decode', d['config']['workload'][:40], '|', d['value'], d['unit'], '| ms/step', d['ms_per_step'])
"
timeout 400 python bench.py --mode sft --dynamic-s2 --micro-batch 1 --steps 3 --warmup 1 2>"$O/val_sft_s2.err" | tee "$O/val_sft_s2.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sft dynamic_s2 ->', d['ms_per_step'], 'ms')"
timeout 400 python bench.py --config nvila_lite_3b --no-sft --no-sustain --no-cpu-baseline 2>"$O/val_bench_lite3b.err" | tail -1 > "$O/val_bench_lite3b.json"
python -c "
import json
d = json.loads(open('gpurun_out/val_bench_lite3b.json').read().strip().splitlines()[-1]); print('lite-3b', d['value'], d['unit'], 'ttft', d.get('ttft_ms'))
"
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
if [ -z "$VAL_LIGHT" ]; then
timeout 600 bash tools/pmc.sh val --no-sft --no-sustain --steps 8 --warmup 2 | tail -2
for CTR in FETCH_SIZE WRITE_SIZE; do
  DB=$(find "$O/pmc_val_$CTR" -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py "$DB" gemv_kernel | head -8 >> "$O/val_pmc_hbm_counters.txt"
  [ -n "$DB" ] && python tools/pmc_summary.py "$DB" gemm256_kernel | head -8 >> "$O/val_pmc_hbm_counters.txt"
done
python tools/pmc_traffic_json.py "$O/val_pmc_hbm_counters.txt" "$O/val_pmc_traffic.json" r06 | cut -c1-200
find "$O" -path "*pmc_val_*" -name "*.db" -delete
# HBM traffic of the contraction-major (wgrad) GEMMs with THIS build's tile order (the round-3 json predated the 8 x 4 patches)
timeout 900 bash tools/pmc_gemm_sft.sh 2>&1 | tail -6
cp "$O/pmc_gemm_sft/summary.txt" "$O/val_pmc_gemm_sft.txt" 2>/dev/null; cp "$O/pmc_gemm_sft/summary.json" "$O/val_pmc_gemm_sft.json" 2>/dev/null
timeout 900 bash tools/pmc_mfma.sh val_sft --mode sft --steps 2 --warmup 1 | tail -1
cp "$O/pmc_mfma_val_sft/summary.txt" "$O/val_pmc_mfma_sft_step.txt" 2>/dev/null
timeout 600 bash tools/pmc_mfma.sh val_ttft --no-sft --no-sustain --steps 8 --warmup 2 | tail -1
cp "$O/pmc_mfma_val_ttft/summary.txt" "$O/val_pmc_mfma_ttft_decode.txt" 2>/dev/null
fi
timeout 600 bash tools/profile.sh val_sft --mode sft --steps 3 --warmup 1 2>&1 | tail -1
if [ -f "$O/prof_val_sft/trace_results.db" ]; then
  python tools/rocpd_summary.py "$O/prof_val_sft/trace_results.db" "$O/val_sft_kernel_stats.csv"; rm -f "$O/prof_val_sft/trace_results.db"
fi
timeout 600 bash tools/profile.sh val --no-sft --no-sustain --steps 32 --warmup 8 2>&1 | tail -1
if [ -f "$O/prof_val/trace_results.db" ]; then
  python tools/rocpd_summary.py "$O/prof_val/trace_results.db" "$O/val_bench_kernel_stats.csv"
  python tools/rocpd_timeline.py "$O/prof_val/trace_results.db" im2col_kernel argmax_stage2 median "$O/val_ttft_timeline.txt"
  rm -f "$O/prof_val/trace_results.db"
fi
find "$O" -name "*.db" -size +1M -delete
[ -z "$VAL_LIGHT" ] && head -8 "$O/val_pmc_mfma_sft_step.txt"; true
