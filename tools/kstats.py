#!/usr/bin/env python
"""Print a compact per-kernel table from a rocprofv3 *_kernel_stats.csv."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'].split('(')[0][:44].ljust(46), r['Calls'].rjust(6), ('%.2f' % (float(r['AverageNs']) / 1e3)).rjust(10), 'us', r['Percentage'])
