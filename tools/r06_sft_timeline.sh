#!/bin/bash
# One SFT step as a timeline: per-stream busy time and the largest gaps (im2col = start of a step).  usage (GPU box): bash tools/r06_sft_timeline.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_sft_tl
timeout 600 rocprofv3 --kernel-trace -d $O/prof_sft_tl -o trace -- python $REPO/bench.py --mode sft --steps 4 --warmup 2 > $O/prof_sft_tl.log 2>&1
DB=$(find $O/prof_sft_tl -name "*.db" | head -1)
cd $REPO && python tools/rocpd_timeline.py "$DB" im2col_kernel @next -2 $O/r06_sft_step_timeline.txt
python - "$DB" >> $O/r06_sft_step_timeline.txt <<'PY'
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
st = [i for i, r in enumerate(rows) if "im2col_kernel" in r[0]]
a, b = st[-2], st[-1]
t0, t1 = rows[a][1], rows[b][1]
# idle intervals per stream inside the step, and which kernels bound them
per = defaultdict(list)
for n, s, e, sid in rows[a:b]:
    per[sid].append((s, e, n))
print("\nper stream: first start, last end (ms into the step), busy ms, kernels; largest idle gaps")
for sid, ks in sorted(per.items()):
    busy = sum(e - s for s, e, _ in ks) / 1e6
    print(f"stream {sid}: {(ks[0][0] - t0) / 1e6:8.2f} .. {(ks[-1][1] - t0) / 1e6:8.2f}  busy {busy:8.2f}  n={len(ks)}")
    gaps = sorted(((ks[i + 1][0] - ks[i][1], ks[i][2][:50], ks[i + 1][2][:50], (ks[i][1] - t0) / 1e6) for i in range(len(ks) - 1)), reverse=True)[:6]
    for g, x, y, at in gaps:
        print(f"      gap {g / 1e3:9.1f} us at {at:7.2f} ms: {x} -> {y}")
print(f"step wall {(t1 - t0) / 1e6:.2f} ms")
PY
find $O/prof_sft_tl -name "*.db" -delete
tail -40 $O/r06_sft_step_timeline.txt
