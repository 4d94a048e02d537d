#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name (sum over dispatches):
usage: python tools/pmc_by_kernel.py <counter_collection.csv> [min_share]
Prints, per kernel, dispatch count, total duration and every counter's sum, plus MFMA-busy / (SIMDs x time) when available."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
seen = set()
with open(path) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].replace("void ", "").replace("w2l::", "").split("(")[0]
        key = (name, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            agg[name]["dispatches"] += 1
            agg[name]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
tot = sum(v["ns"] for v in agg.values())
print("%-58s %6s %10s %6s  counters" % ("kernel", "calls", "ms", "%"))
for name, v in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    if v["ns"] < 0.005 * tot:
        continue
    extra = " ".join("%s=%.4g" % (k, x) for k, x in sorted(v.items()) if k not in ("dispatches", "ns"))
    line = "%-58s %6d %10.3f %5.1f%%  %s" % (name[:58], v["dispatches"], v["ns"] / 1e6, 100 * v["ns"] / tot, extra)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v["ns"] > 0:
        # busy cycles are summed over the 1024 SIMDs; divide by SIMDs x elapsed cycles at the 2.4 GHz peak clock
        line += "  mfma_busy_frac@2.4GHz=%.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["ns"] * 2.4))
    print(line)
