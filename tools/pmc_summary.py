#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection.csv files: mean counter value per kernel name."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r"(nw_[a-z_0-9]+(<[^>]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} mean={sum(v)/len(v):16.1f}  n={len(v)}")
