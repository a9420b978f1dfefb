#!/usr/bin/env python
"""Timeline of the iteration kernels from a rocprofv3 *_kernel_trace.csv: mean duration per kernel and mean gap to the next kernel."""
import csv, sys, collections
rows = sorted((r for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "lili::" in r["Kernel_Name"]]
rows = rows[len(rows) // 2:]          # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    k = a["Kernel_Name"].split("(")[0].replace("void ", "")
    dur[k].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[k].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k in dur:
    print(f"{k:42s} n={len(dur[k]):5d} dur {sum(dur[k]) / len(dur[k]) / 1e3:7.2f} us   gap-after {sum(gap[k]) / len(gap[k]) / 1e3:6.2f} us")
