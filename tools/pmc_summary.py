#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel dispatch for ONE bench step (the dispatches between
the last two w2l::datagen_pack_kernel launches = one full pass of the hot path).
usage: python tools/pmc_summary.py gpurun_out/<tag>/prof_pmc*/pmc*_counter_collection.csv

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM section), so the fetch column below is already doubled ("fetch_MB_x2")."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(dict)   # (run, dispatch id) -> {counter: value, meta}
for path in sys.argv[1:]:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            key = (path.split("/")[-2], int(r["Dispatch_Id"]))
            d = rows[key]
            d["name"] = r["Kernel_Name"]
            d["grid"] = int(r["Grid_Size"])
            d["wg"] = int(r["Workgroup_Size"])
            d["vgpr"] = r["VGPR_Count"]
            d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            d[r["Counter_Name"]] = float(r["Counter_Value"])

by_run = defaultdict(list)
for (run, did), d in sorted(rows.items()):
    by_run[run].append((did, d))

META = ("name", "grid", "wg", "vgpr", "ns")


def short(name):
    n = name.replace("void ", "").replace("w2l::", "")
    return n.split("(")[0].replace(" ", "")


for run, lst in by_run.items():
    marks = [j for j, (_, d) in enumerate(lst) if "datagen_pack_kernel" in d["name"]]
    if len(marks) >= 2:
        step = lst[marks[-2]:marks[-1]]
    else:
        step = lst
    counters = sorted({k for _, d in step for k in d if k not in META})
    print("== %s: one step = %d dispatches; counters: %s" % (run, len(step), counters))
    print("%3s %-36s %7s %5s %9s " % ("#", "kernel", "blocks", "vgpr", "us") + " ".join("%14s" % c[-14:] for c in counters))
    tot = defaultdict(float)
    for j, (i, d) in enumerate(step):
        print("%3d %-36s %7d %5s %9.1f " % (j, short(d["name"])[:36], d["grid"] // d["wg"], d["vgpr"], d["ns"] / 1e3) +
              " ".join("%14.4g" % d.get(c, float("nan")) for c in counters))
        for c in counters:
            tot[c] += d.get(c, 0.0)
        tot["ns"] += d["ns"]
    print("total us %.1f " % (tot["ns"] / 1e3) + " ".join("%s=%.6g" % (c, tot[c]) for c in counters))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "SQ_BUSY_CYCLES" in tot:
        print("MFMA busy / SQ busy = %.3f" % (tot["SQ_VALU_MFMA_BUSY_CYCLES"] / tot["SQ_BUSY_CYCLES"]))
    if "FETCH_SIZE" in tot:
        print("HBM fetch per step: %.1f MB (FETCH_SIZE KiB x2 gfx950 correction)" % (tot["FETCH_SIZE"] * 2 * 1024 / 1e6))
    if "WRITE_SIZE" in tot:
        print("HBM write per step: %.1f MB (WRITE_SIZE KiB, uncalibrated)" % (tot["WRITE_SIZE"] * 1024 / 1e6))
