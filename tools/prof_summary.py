"""Summarise a rocprofv3 rocpd sqlite (kernel trace) as text: per (kernel, grid) calls / total / avg / min / max (us).

Usage: prof_summary.py <results.db> [name filter] [--by-grid]
"--by-grid" keeps launches of one kernel with different grids apart (e.g. the chain kernel's all-layer launches vs the
notebook-default last-layer-only launches), so the average duration can be compared with bench.py's HIP-event timing.
"""
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
by_grid = "--by-grid" in sys.argv
con = sqlite3.connect(args[0])
cur = con.cursor()
flt = args[1] if len(args) > 1 else ""
grid = ", grid_x / workgroup_x" if by_grid else ", 0"
rows = cur.execute("select name%s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels group by name%s order by sum(end-start) desc" % (grid, grid if by_grid else "")).fetchall()
tot = sum(r[3] for r in rows)
print("%-84s %7s %7s %12s %10s %10s %10s %6s" % ("kernel", "wgs", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for n, g, c, s, a, mn, mx in rows:
    if flt and flt not in n:
        continue
    print("%-84s %7s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n[:84], g if by_grid else "-", c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100 * s / tot))
