"""Summarise rocprofv3 --pmc CSV output per kernel: max/avg counter value per dispatch (KB for *_SIZE counters).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane)
coalesced streaming read -> the `bytes_corrected` column doubles FETCH_SIZE; WRITE_SIZE is taken as is.
Usage: pmc_summary.py <counter_collection.csv> [name filter] > summary.txt
"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for r in rows:
    if flt in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("%-72s %-11s %6s %14s %14s %18s" % ("kernel", "counter", "n", "avg_KB", "max_KB", "max_bytes_corrected"))
for (k, c), v in sorted(agg.items()):
    mx = max(v)
    corr = mx * 1024 * (2 if c == "FETCH_SIZE" else 1)
    print("%-72s %-11s %6d %14.1f %14.1f %18.0f" % (k, c, len(v), sum(v) / len(v), mx, corr))
