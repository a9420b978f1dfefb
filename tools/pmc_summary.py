#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel launch."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if not k.startswith("lili::"):
            continue
        print(path.split("/")[-1][:2], k, " ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in d.items()))
