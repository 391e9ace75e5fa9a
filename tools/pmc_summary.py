#!/usr/bin/env python3
"""Summarises a rocprofv3 --pmc counter_collection.csv: mean of each counter per kernel."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"]
        if filt and filt not in name:
            continue
        acc[name[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
