#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch for kernels matching a substring."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print("%-28s mean %16.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
