#!/usr/bin/env python
"""Averages rocprofv3 --pmc CSV output per kernel.  Usage: tools/pmc_summary.py <outdir> [kernel-substring]"""
import csv, glob, os, sys, collections
out = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "sigmaenv"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if sub not in k: continue
        acc[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
            if extra in row and row[extra] != "": acc[k.split("(")[0]]["~" + extra].append(float(row[extra]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} n={len(v):4d} avg={sum(v)/len(v):.4g}")
