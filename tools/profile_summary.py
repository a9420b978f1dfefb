#!/usr/bin/env python
"""Summarises a tools/make_profiles.sh output directory: per-kernel mean durations (kernel-trace) and PMC means,
and writes pmc_traffic.json with the HBM bytes per launch of each lili kernel:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes a 16-B/lane read stream fetches
(MI355X_MICROARCH.md §HBM), hence the factor 2 on the read side; WRITE_SIZE is taken as is (uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.replace("lili::", "")


stats = {}
f = os.path.join(out, "stats_kernel_stats.csv")
if os.path.exists(f):
    for r in csv.DictReader(open(f)):
        stats[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"]))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p_*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, d in pmc.items():
    if not (k.startswith("k_") or k.startswith("lili")):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = dict(counters=m)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
        e["fetch_kib"] = m["FETCH_SIZE"]; e["write_kib"] = m["WRITE_SIZE"]
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if k in stats:
        e["avg_us"] = stats[k]["avg_us"]
    summary[k if "_t<" in k else k.split("<")[0]] = e      # (k_scan_lookback_t<true> / <false>, k_scatter_t<...>: different kernels)
json.dump(summary, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["pct"])[:12]:
    print(f"{k[:44]:46s} {s['calls']:6d} {s['avg_us']:10.2f} us {s['pct']:6.2f} %")
for k, e in summary.items():
    if "hbm_bytes_per_launch" in e:
        print(f"{k[:40]:42s} HBM {e['hbm_bytes_per_launch'] / 1e6:8.2f} MB/launch  L2 hit {e.get('l2_hit_rate', float('nan')):.3f}")
