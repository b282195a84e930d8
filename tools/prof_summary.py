#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --stats kernel_stats.csv."""
import csv
import sys
for path in sys.argv[1:]:
    print('==', path)
    for r in list(csv.DictReader(open(path)))[:10]:
        print('%-62s calls %5s tot %9.1fus avg %8.2fus %6s%%' % (r['Name'][:62], r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                                               float(r['AverageNs']) / 1e3, r['Percentage'][:6]))
