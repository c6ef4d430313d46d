#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per-kernel mean of every counter (last dispatch of each pass)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if not any(t in k for t in ("chain_f32", "q15", "q7", "w14")):
                continue
            name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c in sorted(d):
        v = d[c]
        print(f"   {c:34s} n={len(v):2d}  mean {sum(v)/len(v):16.1f}  last {v[-1]:16.1f}")
