#!/usr/bin/env python3
"""Development helper: mean of every counter per kernel from a rocprofv3 --pmc counter_collection.csv directory."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:60]
        if "million" not in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name, "launches", len(next(iter(cs.values()))))
    for c, v in sorted(cs.items()):
        v = v[len(v) // 4:]      # skip warm-up launches
        print(f"   {c:28s} {sum(v) / len(v):14.1f}")
