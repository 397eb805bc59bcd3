#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_full_depth.py -m gpu -q -k "nccl or backward_probe or identical_bits" 2>&1 | tail -8
timeout 900 python bench.py > $O/r06_bench_mid.json 2> $O/r06_bench_mid.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_bench_mid.json").read().strip().splitlines()[-1])
print("value", d["value"], "ttft", d["ttft_ms"], "sft", (d.get("sft") or {}).get("ms_per_step"), "rccl", (d.get("sft") or {}).get("rccl_ranks_seen"))
cb=d["cpu_baseline"]; print("cpu_baseline", cb["value"], cb["kind"], cb["cores"], "port", cb.get("port_tokens_per_s"), "live", (cb.get("reference_live") or {}).get("measured"), (cb.get("reference_live") or {}).get("error"))
P
VILA_BENCH_FORCE_DIST=1 VILA_GRAD_EXCHANGE=direct timeout 400 python bench.py --mode sft --steps 3 --warmup 1 2>$O/r06_sft_forcedist_direct.err | tail -1 > $O/r06_sft_forcedist_direct.json
VILA_BENCH_FORCE_DIST=1 timeout 400 python bench.py --mode sft --steps 3 --warmup 1 2>$O/r06_sft_forcedist.err | tail -1 > $O/r06_sft_forcedist.json
python - <<'P'
import json
for n in ("direct",""):
    f="gpurun_out/r06_sft_forcedist"+("_"+n if n else "")+".json"
    try:
        d=json.loads(open(f).read()); print(f, d["ms_per_step"], {k:d.get(k) for k in ("exchange_algo","exchange_bytes","exchange_active","rccl_ranks_seen","rccl_backend")})
    except Exception as e: print(f, "failed", e)
P
