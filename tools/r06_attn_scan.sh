#!/bin/bash
# usage (GPU box): bash tools/r06_attn_scan.sh  -> gpurun_out/r06_attn_scan_ks{0,1}.txt : the attention forward launches of tools/r06_attn_scan.py in order
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for KS in ${KS_LIST:-0 1}; do
  rm -rf $O/prof_attn_scan
  VILA_ATTN_KS=$KS timeout 300 rocprofv3 --kernel-trace -d $O/prof_attn_scan -o trace -- python $REPO/tools/r06_attn_scan.py > $O/prof_attn_scan.log 2>&1
  DB=$(find $O/prof_attn_scan -name "*.db" | head -1)
  python - "$DB" > $O/r06_attn_scan_ks$KS.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [r for r in db.execute("select name, start, end from kernels order by start") if "attn_fwd" in r[0]]
Ts = (16, 64, 128, 256, 512, 769, 1024)
for i in range(0, len(rows), 6):
    grp = rows[i:i + 6]
    d = sorted((e - s) / 1e3 for _, s, e in grp)
    cfg = i // 6
    print(f"{'hd128 causal 28/4' if cfg < 7 else 'hd72 full 16/16':18s} T={Ts[cfg % 7]:5d}: median {d[len(d)//2]:7.2f} us  min {d[0]:7.2f}  {grp[0][0][:48]}")
PY
  find $O/prof_attn_scan -name "*.db" -delete
done
for KS in ${KS_LIST:-0 1}; do echo "== VILA_ATTN_KS=$KS"; cat $O/r06_attn_scan_ks$KS.txt; done
