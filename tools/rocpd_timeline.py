"""Timeline of one phase out of a rocprofv3 --kernel-trace database (rocpd sqlite): where does the wall-clock between two marker
kernels go — kernels or the gaps between them?
usage: python tools/rocpd_timeline.py trace_results.db <first-kernel-substring> <last-kernel-substring> [occurrence=-1 | median] [out.txt]
("median": of all complete first..last windows the one with the median wall time — a host hiccup in one pass does not end up as the timeline)
e.g.   python tools/rocpd_timeline.py gpurun_out/prof_r02/trace_results.db im2col_kernel argmax_stage2 -1 profiles/r02_ttft_timeline.txt
The window runs from the start of the chosen occurrence of the first marker to the end of the next occurrence of the last marker;
with "@next" as the last marker it runs to the start of the NEXT occurrence of the first marker (one period, e.g. one training step).
Per-stream busy time is printed as well (kernels of different streams overlap)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
first, last = sys.argv[2], sys.argv[3]
occ = sys.argv[4] if len(sys.argv) > 4 else "-1"
out = open(sys.argv[5], "w") if len(sys.argv) > 5 else sys.stdout
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
starts = [i for i, r in enumerate(rows) if first in r[0]]
if not starts:
    sys.exit(f"no kernel matching {first!r}")
if occ == "median" and last != "@next":
    cands = []
    for a in starts:
        b = next((i for i in range(a, len(rows)) if last in rows[i][0]), None)
        nxt = next((x for x in starts if x > a), len(rows))
        if b is not None and b < nxt:                           # a complete window with no second start marker inside
            cands.append((rows[b][2] - rows[a][1], a))
    if not cands:
        sys.exit("no complete window")
    cands.sort()
    i0 = cands[len(cands) // 2][1]
else:
    i0 = starts[int(occ)]
if last == "@next":
    k = starts.index(i0)
    if k + 1 >= len(starts):
        sys.exit("no later occurrence of the first marker")
    i1 = starts[k + 1] - 1
else:
    i1 = next((i for i in range(i0, len(rows)) if last in rows[i][0]), None)
    if i1 is None:
        sys.exit(f"no kernel matching {last!r} after the marker")
win = rows[i0:i1 + 1]
t0, t1 = win[0][1], (rows[i1 + 1][1] if last == "@next" else win[-1][2])
busy, cur_end, gaps = 0, t0, []
per = defaultdict(lambda: [0, 0])
for k, (name, s, e, _) in enumerate(win):
    per[name][0] += 1
    per[name][1] += e - s
    if s > cur_end:
        gaps.append((s - cur_end, win[k - 1][0] if k else "", name))
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
tot = t1 - t0
print(f"window: {len(win)} kernels, {tot / 1e3:.1f} us wall; busy (union of kernels) {busy / 1e3:.1f} us = {100 * busy / tot:.1f} %; "
      f"gaps {sum(g[0] for g in gaps) / 1e3:.1f} us in {len(gaps)} gaps", file=out)
per_stream = defaultdict(int)
for name, s_, e_, st in win:
    per_stream[st] += e_ - s_
print("per stream busy (us): " + ", ".join(f"stream {k}: {v / 1e3:.1f}" for k, v in sorted(per_stream.items(), key=lambda kv: -kv[1])), file=out)
print("\nper kernel (calls, total us, avg us):", file=out)
for name, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {c:5d} {d / 1e3:10.1f} {d / c / 1e3:9.2f}  {name[:120]}", file=out)
print("\nlargest gaps (us, after -> before):", file=out)
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"  {g / 1e3:8.1f}  {a[:60]} -> {b[:60]}", file=out)
hist = defaultdict(int)
for g, _, _ in gaps:
    hist[min(int(g / 1e3), 20)] += 1
print("\ngap histogram (us bucket: count): " + ", ".join(f"{k}{'+' if k == 20 else ''}: {v}" for k, v in sorted(hist.items())), file=out)
