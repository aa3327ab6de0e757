#!/usr/bin/env python3
import csv, glob, sys, collections
f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:40s} {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
