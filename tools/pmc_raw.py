"""Raw rocprofv3 --pmc counters per (kernel, grid size): mean per dispatch, plus SQ ratios when those counters exist.

Usage: pmc_raw.py <counter_collection.csv> [name filter]
"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"]
    if flt not in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", name.split("(mmx::")[0])[-60:]
    agg[(short, r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print("%s  grid=%s  n=%d" % (k, grid, len(next(iter(cs.values())))))
    for c, v in sorted(m.items()):
        print("    %-28s %16.0f" % (c, v))
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                print("    %-28s %15.1f%% of wave cycles" % (c, 100 * m[c] / w))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        print("    MFMA busy / SQ busy          %15.2f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"]))
