#!/bin/bash
# Per-launch rows of one TTFT at the SHORT prompt (1 image + 32 tokens, S = 289).  usage (GPU box): bash tools/r06_ttft_rows_short.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_rows_s
timeout 400 rocprofv3 --kernel-trace -d $O/prof_rows_s -o trace -- python $REPO/bench.py --prompt-tokens 32 --no-cpu-baseline --no-sft --no-sustain --steps 8 --warmup 2 > $O/prof_rows_s.log 2>&1
DB=$(find $O/prof_rows_s -name "*.db" | head -1)
cd $REPO && python tools/rocpd_rows.py "$DB" im2col_kernel argmax_stage2 -2 $O/r06_ttft_rows_s289.txt "attn_fwd_kernel<128"
find $O/prof_rows_s -name "*.db" -delete
python - <<'PY'
import re
rows = [l.split(None, 4) for l in open("gpurun_out/r06_ttft_rows_s289.txt").read().splitlines()[1:]]
agg = {}
for s, e, d, st, n in rows:
    k = n[:64]; c = agg.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += float(d)
print(f"{len(rows)} launches, {float(rows[-1][1]):.1f} us")
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
    print(f"{c:4d} x {t / c:7.2f} = {t:8.1f} us  {k}")
PY
