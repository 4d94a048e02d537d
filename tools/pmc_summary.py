#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel dispatch (last bench step only).
usage: python tools/pmc_summary.py gpurun_out/<tag>/prof_pmc*/pmc*_counter_collection.csv"""
import csv
import sys
from collections import OrderedDict, defaultdict

rows = defaultdict(dict)   # dispatch id -> {counter: value, meta}
for path in sys.argv[1:]:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            key = (path.split("/")[-2], int(r["Dispatch_Id"]))
            d = rows[key]
            d["name"] = r["Kernel_Name"]
            d["grid"] = int(r["Grid_Size"])
            d["wg"] = int(r["Workgroup_Size"])
            d["vgpr"] = r["VGPR_Count"]
            d["lds"] = r["LDS_Block_Size"]
            d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            d[r["Counter_Name"]] = float(r["Counter_Value"])

by_run = defaultdict(list)
for (run, did), d in sorted(rows.items()):
    by_run[run].append((did, d))

for run, lst in by_run.items():
    conv = [(i, d) for i, d in lst if "conv_igemm" in d["name"]]
    n_per_step = 51
    last = conv[-n_per_step:]
    counters = [k for k in last[0][1] if k not in ("name", "grid", "wg", "vgpr", "lds", "ns")]
    print("== %s: %d conv dispatches, showing the last %d; counters: %s" % (run, len(conv), len(last), counters))
    print("%3s %-14s %7s %5s %9s " % ("#", "tile", "blocks", "vgpr", "us") + " ".join("%14s" % c[-14:] for c in counters))
    tot = defaultdict(float)
    for j, (i, d) in enumerate(last):
        tile = d["name"].split("<")[1].split(">")[0].replace(" ", "")
        print("%3d %-14s %7d %5s %9.1f " % (j, tile, d["grid"] // d["wg"], d["vgpr"], d["ns"] / 1e3) +
              " ".join("%14.4g" % d.get(c, float("nan")) for c in counters))
        for c in counters:
            tot[c] += d.get(c, 0.0)
        tot["ns"] += d["ns"]
    print("total us %.1f " % (tot["ns"] / 1e3) + " ".join("%s=%.4g" % (c, tot[c]) for c in counters))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "SQ_BUSY_CYCLES" in tot:
        print("MFMA busy / SQ busy = %.3f" % (tot["SQ_VALU_MFMA_BUSY_CYCLES"] / tot["SQ_BUSY_CYCLES"]))
