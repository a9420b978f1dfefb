#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel launch.
--last N: only the last N launches of each kernel (rows are in dispatch order)."""
import csv, sys, collections
args = sys.argv[1:]
last = 0
if "--last" in args:
    i = args.index("--last")
    last = int(args[i + 1])
    del args[i:i + 2]
for path in args:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if not k.startswith("lili::"):
            continue
        if last:
            d = {c: v[-last:] for c, v in d.items()}
        print(path.split("/")[-1][:2], k, " ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in d.items()))
