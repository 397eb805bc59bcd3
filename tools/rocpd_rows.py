"""Per-kernel rows (start offset, duration, stream, name) of one window out of a rocprofv3 --kernel-trace database: the window from the
chosen occurrence of <first> to the next <last>.  usage: python tools/rocpd_rows.py trace_results.db <first> <last> [occurrence=-2] [out.txt] [must_contain]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first, last = sys.argv[2], sys.argv[3]
occ = int(sys.argv[4]) if len(sys.argv) > 4 else -2
out = open(sys.argv[5], "w") if len(sys.argv) > 5 else sys.stdout
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
starts = [i for i, r in enumerate(rows) if first in r[0]]
must = sys.argv[6] if len(sys.argv) > 6 else ""           # optional: the window must contain this kernel (picks the shortest such window)
if must:
    best = None
    for a in starts:
        b = next((i for i in range(a, len(rows)) if last in rows[i][0]), None)
        if b is None or not any(must in r[0] for r in rows[a:b]):
            continue
        span = rows[b][2] - rows[a][1]
        if best is None or span < best[0]:
            best = (span, a, b)
    _, i0, i1 = best
else:
    i0 = starts[occ]
    i1 = next(i for i in range(i0, len(rows)) if last in rows[i][0])
t0 = rows[i0][1]
print("start_us  end_us  dur_us  stream  kernel", file=out)
for name, s, e, st in rows[i0:i1 + 1]:
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f}  {st}  {name[:70]}", file=out)
