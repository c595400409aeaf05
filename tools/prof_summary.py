"""Summarise a rocprofv3 rocpd sqlite (kernel trace) as text: per-kernel calls / total / avg / min / max (us)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows)
print("%-90s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for n, c, s, a, mn, mx in rows:
    if flt and flt not in n:
        continue
    print("%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n[:90], c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100 * s / tot))
