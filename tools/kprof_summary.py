#!/usr/bin/env python
"""Summarise tools/kprof.sh output: per counter, the value of the LAST dispatch of the biggest conv kernel."""
import csv
import glob
import sys

out = sys.argv[1]
vals = {}
meta = None
for path in sorted(glob.glob(out + "/p*/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(path)) if "conv_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            vals[r["Counter_Name"]] = float(r["Counter_Value"])
            meta = (r["Kernel_Name"][:60], int(r["Grid_Size"]) // int(r["Workgroup_Size"]),
                    int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
print("kernel %s  blocks %d  %.1f us  vgpr %s agpr %s lds %s" % meta)
for k in sorted(vals):
    print("  %-40s %14.5g" % (k, vals[k]))
w = vals.get("SQ_WAVE_CYCLES")
if w:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in vals:
            print("  %s / SQ_WAVE_CYCLES = %.3f" % (k, vals[k] / w))
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals:
    print("  MFMA_BUSY / SQ_BUSY = %.3f" % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / vals["SQ_BUSY_CYCLES"]))
