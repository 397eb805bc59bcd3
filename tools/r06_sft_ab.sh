#!/bin/bash
# Same-box A/B of one environment switch on the SFT step.  usage (GPU box): bash tools/r06_sft_ab.sh VAR [rounds=2] [A=0] [B=1]
VAR=$1; R=${2:-2}; A=${3:-0}; Bv=${4:-1}
for i in $(seq 1 $R); do
  for v in $A $Bv; do
    env $VAR=$v timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v sft', d['ms_per_step'], 'ms  loss', d['loss'])"
  done
done
