"""Per-kernel average of a PMC counter from rocprofv3's rocpd sqlite output.
usage: python tools/pmc_summary.py <db> [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
print("columns:", cols)
name_col = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
q = f"select {name_col}, counter_name, count(*), avg(value), sum(value) from counters_collection group by {name_col}, counter_name order by sum(value) desc"
for name, ctr, n, avg, tot in cur.execute(q):
    if pat in name:
        print(f"{name[:70]:70s} {ctr:12s} n={n:6d} avg={avg:14.1f} sum={tot:16.1f}")
