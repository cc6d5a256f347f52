#!/usr/bin/env python3
"""Per-kernel averages of every counter in rocprofv3 counter_collection csv files.
usage: pmc_summary.py file.csv [file.csv ...]"""
import csv, collections, sys
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        tab[name.split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in tab.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-22s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
