#!/bin/bash
# Kernel stats + one decode token's launches of the W4A16 step.  usage (GPU box): bash tools/r06_w4_rows.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_w4
timeout 400 rocprofv3 --kernel-trace -d $O/prof_w4 -o trace -- python $REPO/bench.py --w4 --no-cpu-baseline --no-sft --no-sustain --steps 16 --warmup 4 > $O/prof_w4.log 2>&1
DB=$(find $O/prof_w4 -name "*.db" | head -1)
cd $REPO && python - "$DB" > $O/r06_w4_token_rows.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
# last complete token: from the last-but-one decode_prologue to the next one
idx = [i for i, r in enumerate(rows) if "decode_prologue" in r[0]]
a, b = idx[-3], idx[-2]
t0 = rows[a][1]
agg = {}
for n, s, e in rows[a:b]:
    k = n[:60]
    c = agg.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += (e - s) / 1e3
print(f"one W4A16 decode token: {b - a} launches, {(rows[b][1] - t0) / 1e3:.1f} us wall")
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{c:4d} x {t / c:7.2f} us = {t:8.1f} us  {k}")
print()
for n, s, e in rows[a:a + 12]:
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f} {(e - s) / 1e3:7.2f}  {n[:70]}")
PY
find $O/prof_w4 -name "*.db" -delete
cat $O/r06_w4_token_rows.txt
