"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.
usage: python tools/rocpd_summary.py gpurun_out/prof_<tag>/trace_results.db profiles/<name>.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for name, calls, tot, avg, pct in rows:
        if len(name) > 160:
            name = name[:157] + "..."
        w.writerow([name, calls, f"{tot:.3f}", f"{avg:.3f}", f"{pct:.3f}"])
print(f"{len(rows)} kernels -> {sys.argv[2]}")
