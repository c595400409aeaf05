"""Per-kernel averages of SQ counters from a rocprofv3 --pmc CSV, as fractions of SQ_WAVE_CYCLES where that makes sense.
Usage: pmc_sq.py <counter_collection.csv> [name filter]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if flt in r["Kernel_Name"]:
        name = r["Kernel_Name"].replace("void mmx::(anonymous namespace)::", "").split("(")[0][:60]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    print(k, "(%d dispatches)" % len(next(iter(cs.values()))))
    for c, v in sorted(avg.items()):
        print("    %-26s %16.0f %s" % (c, v, ("%6.1f %% of wave cycles" % (100 * v / wc)) if wc and c != "SQ_WAVE_CYCLES" else ""))
