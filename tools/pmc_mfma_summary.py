"""Per-kernel MFMA utilisation from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES pass (rocpd sqlite).
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
  rocprofv3 sums every counter over its instances: SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs of the chip, GRBM_GUI_ACTIVE over the 8 XCDs (so /8 gives
  the kernel's active cycles: 658 k cycles = 274 us for the 258-us gemm256 launches, and busy cycles / 16 per 16x16x32 MFMA x 16384 flop reproduces the
  kernel's TFLOP/s).  1.0 = every SIMD's matrix pipe busy every cycle = the 2.5 PFLOP/s dense bf16 peak.
usage: python tools/pmc_mfma_summary.py <db>"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
name_col = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for name, ctr, n, tot in cur.execute(f"select {name_col}, counter_name, count(*), sum(value) from counters_collection group by {name_col}, counter_name"):
    agg[name][ctr] += tot
    calls[name] = max(calls[name], n)
rows = []
for name, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    util = mf / (gui / 8 * 256 * 4) if gui else 0.0
    rows.append((mf, name, calls[name], gui, util, c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_WAVES", 0.0)))
rows.sort(reverse=True)
print(f"{'kernel':72s} {'calls':>6s} {'GUI_ACTIVE':>14s} {'MFMA_BUSY':>16s} {'MfmaUtil':>9s}")
tg = tm = 0.0
for mf, name, n, gui, util, sqb, wv in rows[:40]:
    print(f"{name[:72]:72s} {n:6d} {gui:14.0f} {mf:16.0f} {util:9.3f}")
for mf, name, n, gui, util, sqb, wv in rows:
    tg += gui; tm += mf
print(f"{'ALL KERNELS':72s} {'':6s} {tg:14.0f} {tm:16.0f} {tm / (tg / 8 * 1024) if tg else 0:9.3f}")
