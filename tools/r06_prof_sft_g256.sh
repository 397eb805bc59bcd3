#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
VILA_SFT_ADAMW_GRID=256 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_sft_g256 -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode sft --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof_sft_g256 -name "*.db" | head -1)
python tools/rocpd_summary.py $DB $O/r06_sft_g256_kernel_stats.csv
python - "$DB" <<'P'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t.lower()]
print(kd[:5])
P
find $O/prof_sft_g256 -name "*.db" -delete
