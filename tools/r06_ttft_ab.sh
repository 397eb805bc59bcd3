#!/bin/bash
# Same-box A/B of one environment switch on the TTFT.  usage (GPU box): bash tools/r06_ttft_ab.sh VAR [rounds=2] [A=0] [B=1]
VAR=$1; R=${2:-2}; A=${3:-0}; Bv=${4:-1}
O=gpurun_out; mkdir -p $O
for i in $(seq 1 $R); do
  for v in $A $Bv; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-sft --no-sustain --steps 32 --warmup 8 2>/dev/null | tail -1 > $O/ab_tmp.json
    python - "$VAR=$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read())
print(f"{sys.argv[1]}: ttft {d['ttft_ms']:.3f} ms, decode {d['value']:.1f} tok/s")
PY
  done
done
