"""Per-kernel durations from a rocprofv3 rocpd sqlite, names shortened so that template arguments stay visible (median / min / avg us).
Usage: prof_kernels.py <results.db> [name filter]"""
import sqlite3, sys, statistics, collections
con = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for name, d in con.execute("select name, end-start from kernels"):
    if flt in name:
        short = name.replace("void mmx::(anonymous namespace)::", "").replace("mmx::", "").split("(")[0]
        agg[short].append(d / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-60s calls %4d  median %9.2f  min %9.2f  avg %9.2f us" % (k[:60], len(v), statistics.median(v), min(v), sum(v) / len(v)))
